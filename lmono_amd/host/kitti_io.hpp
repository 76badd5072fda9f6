// lmono_amd/host/kitti_io.hpp -- on-disk formats either side of the hot path.
//   in : KITTI odometry layout -- velodyne/%06d.bin (float32 x y z reflectance), times.txt (one stamp per line),
//        poses/XX.txt (12 doubles per line: row-major 3x4 [R | t])                      (upstream kittiHelper)
//   out: trajectory lines "stamp x y z qx qy qz qw" exactly as the reference prints them
//        (mono_lidar_mapping/src/image_process/Estimator.cc:270-271 "loam_odometry", :642-643 "new_odometry") and the
//        timing log "stamp track_time laser_decode_time pred_time" (Estimator.cc:647)
//        the colour map "rgb_map<index>.ply" of MapBuilder::processMapping (Map_Builder.cc:72-77, pcl::io::savePLYFileBinary
//        of a PointXYZRGB cloud) and the mapping timing log "stamp toc" (map_build_node.cc:230)
#pragma once
#include <array>
#include <cstdio>
#include <string>
#include <vector>

namespace lmono_host {

// appends the scan's points (x y z reflectance) to xyzi; returns the number of points, -1 on I/O error
long read_velodyne_bin(const std::string &path, std::vector<float> &xyzi);
bool read_times(const std::string &path, std::vector<double> &stamps);
bool read_kitti_poses(const std::string &path, std::vector<std::array<double, 12>> &poses);
std::string velodyne_path(const std::string &sequence_dir, int index);   // <dir>/velodyne/%06d.bin

class TrajectoryWriter {
public:
    // style 0: "%f %f %f %f %f %f %f %f\n"  (new_odometry, Estimator.cc:642)
    // style 1: "%f %f %f %f %f %f %f %f \n" (loam_odometry, Estimator.cc:270 -- trailing blank)
    TrajectoryWriter(const std::string &path, int style);
    ~TrajectoryWriter();
    bool ok() const { return f_ != nullptr; }
    // q = (x, y, z, w); the line is flushed like the reference does
    void write(double stamp, const double p[3], const double q_xyzw[4]);
private:
    FILE *f_;
    int style_;
};

class TimingLog {
public:
    explicit TimingLog(const std::string &path);
    ~TimingLog();
    bool ok() const { return f_ != nullptr; }
    void write(double stamp, double track_time, double laser_decode_time, double pred_time);   // Estimator.cc:647
private:
    FILE *f_;
};

// Binary little-endian PLY in the layout pcl::PLYWriter emits for an unorganised pcl::PointCloud<pcl::PointXYZRGB>
// (PCL is a system dependency of the reference, absent here: layout restated from PCL's writer -- vertex element
// x y z float + red green blue uchar, followed by PCL's one-record "camera" element).  pts: [n] records of
// { float x, y, z; uint32 b | g << 8 | r << 16 | a << 24 } (lmono_point_rgb).
struct PointRgb { float x, y, z; unsigned int bgra; };
bool write_ply_binary(const std::string &path, const PointRgb *pts, size_t n);
bool read_ply_binary(const std::string &path, std::vector<PointRgb> &pts);     // reads what write_ply_binary wrote
std::string rgb_map_path(const std::string &dir, int map_index);               // <dir>/rgb_map<index>.ply (Map_Builder.cc:75)

class MappingLog {                                                             // map_build_node.cc:230 "%f %f \n"
public:
    explicit MappingLog(const std::string &path);
    ~MappingLog();
    bool ok() const { return f_ != nullptr; }
    void write(double stamp, double toc_ms);
private:
    FILE *f_;
};

} // namespace lmono_host
