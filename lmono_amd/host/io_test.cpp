// lmono_amd/host/io_test.cpp -- file-format round trip driver (tests/test_kitti_io.py): reads a KITTI-layout sequence
// directory, echoes what it found, and writes the trajectory / timing files in the reference's formats.
//   io_test <sequence_dir> <poses_file> <out_dir> [points.bin]
// With points.bin ([n] records x y z float32 + bgra uint32) it also writes <out_dir>/rgb_map10.ply (the colour map format
// of MapBuilder::processMapping), reads it back and writes <out_dir>/mapping_recorder.txt.
#include "kitti_io.hpp"

#include <cmath>
#include <cstdio>

using namespace lmono_host;

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const std::string seq = argv[1], poses_file = argv[2], out = argv[3];
    std::vector<double> stamps;
    if (!read_times(seq + "/times.txt", stamps)) { std::fprintf(stderr, "times.txt missing\n"); return 1; }
    std::vector<std::array<double, 12>> poses;
    if (!read_kitti_poses(poses_file, poses)) { std::fprintf(stderr, "poses unreadable\n"); return 1; }
    std::printf("STAMPS %zu POSES %zu\n", stamps.size(), poses.size());
    TrajectoryWriter new_odometry(out + "/new_odometry.txt", 0), loam_odometry(out + "/loam_odometry.txt", 1);
    TimingLog times(out + "/times_recorder.txt");
    if (!new_odometry.ok() || !loam_odometry.ok() || !times.ok()) return 1;
    for (size_t k = 0; k < stamps.size(); k++) {
        std::vector<float> xyzi;
        const long n = read_velodyne_bin(velodyne_path(seq, (int)k), xyzi);
        double sum = 0;
        for (float v : xyzi) sum += (double)v;
        std::printf("SCAN %zu %ld %.9g\n", k, n, sum);
        if (k < poses.size()) {
            // translation column of [R | t]; identity rotation as a quaternion is enough for the format check
            const double p[3] = { poses[k][3], poses[k][7], poses[k][11] };
            const double q[4] = { 0.0, 0.0, std::sin(0.05 * (double)k), std::cos(0.05 * (double)k) };
            new_odometry.write(stamps[k], p, q);
            loam_odometry.write(stamps[k], p, q);
            times.write(stamps[k], 0.001 * (double)k, 0.002, 0.5 + (double)k);
        }
    }
    if (argc > 4) {
        FILE *f = std::fopen(argv[4], "rb");
        if (!f) return 1;
        std::vector<PointRgb> pts, back;
        PointRgb p;
        while (std::fread(&p, sizeof p, 1, f) == 1) pts.push_back(p);
        std::fclose(f);
        const std::string ply = rgb_map_path(out, 10);
        if (!write_ply_binary(ply, pts.data(), pts.size()) || !read_ply_binary(ply, back) || back.size() != pts.size()) return 1;
        size_t same = 0;
        for (size_t i = 0; i < pts.size(); i++) same += back[i].x == pts[i].x && back[i].y == pts[i].y && back[i].z == pts[i].z && (back[i].bgra & 0xffffffu) == (pts[i].bgra & 0xffffffu);
        std::printf("PLY %zu %zu\n", pts.size(), same);
        MappingLog rec(out + "/mapping_recorder.txt");
        rec.write(1.5, 2.25); rec.write(2.5, 0.125);
    }
    // a file that is not a whole number of 16-byte records is rejected
    std::vector<float> bad;
    std::printf("BAD %ld\n", read_velodyne_bin(seq + "/times.txt", bad));
    return 0;
}
