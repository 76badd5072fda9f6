"""Synthetic input generators (S1 scans, S2 BA windows): input plumbing for tests and bench.py."""
