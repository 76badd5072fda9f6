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

// ---- PLY (pcl::io::savePLYFileBinary of a PointXYZRGB cloud) ----
static const char *kPlyCameraProps[] = { "view_px", "view_py", "view_pz", "x_axisx", "x_axisy", "x_axisz", "y_axisx", "y_axisy", "y_axisz",
                                          "z_axisx", "z_axisy", "z_axisz", "focal", "scalex", "scaley", "centerx", "centery" };

bool write_ply_binary(const std::string &path, const PointRgb *pts, size_t n)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "ply\nformat binary_little_endian 1.0\ncomment PCL generated\nelement vertex %zu\n", n);
    std::fprintf(f, "property float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n");
    std::fprintf(f, "element camera 1\n");
    for (const char *p : kPlyCameraProps) std::fprintf(f, "property float %s\n", p);
    std::fprintf(f, "property int viewportx\nproperty int viewporty\nproperty float k1\nproperty float k2\nend_header\n");
    std::vector<unsigned char> buf(n * 15);
    for (size_t i = 0; i < n; i++) {
        unsigned char *o = buf.data() + i * 15;
        std::memcpy(o, &pts[i].x, 12);
        o[12] = (unsigned char)(pts[i].bgra >> 16); o[13] = (unsigned char)(pts[i].bgra >> 8); o[14] = (unsigned char)pts[i].bgra;
    }
    bool ok = n == 0 || std::fwrite(buf.data(), 15, n, f) == n;
    // camera record: sensor origin 0, identity axes, focal / scale / centre 0, viewport = (width, height) of the cloud, k1 k2 0
    float cam[17] = { 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0 };
    int viewport[2] = { (int)n, 1 };
    float k[2] = { 0, 0 };
    ok = ok && std::fwrite(cam, 4, 17, f) == 17 && std::fwrite(viewport, 4, 2, f) == 2 && std::fwrite(k, 4, 2, f) == 2;
    return std::fclose(f) == 0 && ok;
}

bool read_ply_binary(const std::string &path, std::vector<PointRgb> &pts)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char line[256];
    size_t n = 0;
    bool header = false, binary = false;
    while (std::fgets(line, sizeof line, f)) {
        if (std::strncmp(line, "format binary_little_endian", 27) == 0) binary = true;
        if (std::sscanf(line, "element vertex %zu", &n) == 1) continue;
        if (std::strncmp(line, "end_header", 10) == 0) { header = true; break; }
    }
    if (!header || !binary) { std::fclose(f); return false; }
    std::vector<unsigned char> buf(n * 15);
    if (n > 0 && std::fread(buf.data(), 15, n, f) != n) { std::fclose(f); return false; }
    std::fclose(f);
    pts.resize(n);
    for (size_t i = 0; i < n; i++) {
        const unsigned char *o = buf.data() + i * 15;
        std::memcpy(&pts[i].x, o, 12);
        pts[i].bgra = (unsigned int)o[14] | (unsigned int)o[13] << 8 | (unsigned int)o[12] << 16 | 0xff000000u;
    }
    return true;
}

std::string rgb_map_path(const std::string &dir, int map_index) { return dir + "/rgb_map" + std::to_string(map_index) + ".ply"; }

MappingLog::MappingLog(const std::string &path) : f_(std::fopen(path.c_str(), "w")) {}
MappingLog::~MappingLog() { if (f_) std::fclose(f_); }
void MappingLog::write(double stamp, double toc_ms)
{
    if (!f_) return;
    std::fprintf(f_, "%f %f \n", stamp, toc_ms);
    std::fflush(f_);
}

} // namespace lmono_host
