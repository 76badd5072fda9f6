// lmono_amd/host/kitti_io.cpp -- see kitti_io.hpp
#include "kitti_io.hpp"

#include <cstring>

namespace lmono_host {

long read_velodyne_bin(const std::string &path, std::vector<float> &xyzi)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return -1;
    if (std::fseek(f, 0, SEEK_END) != 0) { std::fclose(f); return -1; }
    const long bytes = std::ftell(f);
    std::rewind(f);
    if (bytes < 0 || bytes % 16 != 0) { std::fclose(f); return -1; }   // whole float32 x y z reflectance records only
    const size_t n = (size_t)bytes / 16, at = xyzi.size();
    xyzi.resize(at + 4 * n);
    const size_t got = n ? std::fread(xyzi.data() + at, 16, n, f) : 0;
    std::fclose(f);
    if (got != n) { xyzi.resize(at); return -1; }
    return (long)n;
}

bool read_times(const std::string &path, std::vector<double> &stamps)
{
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    double t;
    while (std::fscanf(f, "%lf", &t) == 1) stamps.push_back(t);
    std::fclose(f);
    return true;
}

bool read_kitti_poses(const std::string &path, std::vector<std::array<double, 12>> &poses)
{
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    std::array<double, 12> p;
    for (;;) {
        int k = 0;
        for (; k < 12; k++) if (std::fscanf(f, "%lf", &p[(size_t)k]) != 1) break;
        if (k < 12) { std::fclose(f); return k == 0; }   // a truncated last line is an error
        poses.push_back(p);
    }
}

std::string velodyne_path(const std::string &sequence_dir, int index)
{
    char name[32];
    std::snprintf(name, sizeof(name), "%06d.bin", index);
    return sequence_dir + "/velodyne/" + name;
}

TrajectoryWriter::TrajectoryWriter(const std::string &path, int style) : f_(std::fopen(path.c_str(), "w")), style_(style) {}
TrajectoryWriter::~TrajectoryWriter() { if (f_) std::fclose(f_); }
void TrajectoryWriter::write(double stamp, const double p[3], const double q[4])
{
    if (!f_) return;
    std::fprintf(f_, style_ == 1 ? "%f %f %f %f %f %f %f %f \n" : "%f %f %f %f %f %f %f %f\n", stamp, p[0], p[1], p[2], q[0], q[1], q[2], q[3]);
    std::fflush(f_);
}

TimingLog::TimingLog(const std::string &path) : f_(std::fopen(path.c_str(), "w")) {}
TimingLog::~TimingLog() { if (f_) std::fclose(f_); }
void TimingLog::write(double stamp, double track_time, double laser_decode_time, double pred_time)
{
    if (!f_) return;
    std::fprintf(f_, "%f %f %f %f\n", stamp, track_time, laser_decode_time, pred_time);
    std::fflush(f_);
}

} // namespace lmono_host
