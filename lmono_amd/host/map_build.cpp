// lmono_amd/host/map_build.cpp -- the map_builder node's loop (mono_lidar_mapping/src/map_build_node.cc:140-236) over files:
// for every frame k: scan <seq>/velodyne/%06d.bin + image <seq>/image_bgr/%06d.bgr (raw BGR8, width * height * 3 bytes: this
// image has no PNG decoder) + pose line k of a trajectory file ("stamp x y z qx qy qz qw", the format the estimator writes)
// -> MapBuilder::associateToMap -> processMapping; writes <out>/rgb_map<index>.ply every 10 frames and the timing log
// <out>/mapping_recorder.txt, and dumps the products of the last frame (<out>/depth_last.u8, <out>/cloud_cam_last.bin,
// <out>/cloud_world_last.bin) for inspection.
//   map_build <sequence_dir> <trajectory.txt> <out_dir> [n_frames] [width height fx fy cx cy]
#include "kitti_io.hpp"
#include "lmono_host.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace lmono_host;

static bool read_file(const std::string &path, std::vector<uint8_t> &out, size_t expect)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    out.resize(expect);
    const bool ok = std::fread(out.data(), 1, expect, f) == expect && std::fgetc(f) == EOF;
    std::fclose(f);
    return ok;
}

template <typename T> static bool dump(const std::string &path, const std::vector<T> &v)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = v.empty() || std::fwrite(v.data(), sizeof(T), v.size(), f) == v.size();
    return std::fclose(f) == 0 && ok;
}

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: map_build <sequence_dir> <trajectory.txt> <out_dir> [n_frames] [width height fx fy cx cy]\n"); return 2; }
    const std::string seq = argv[1], traj = argv[2], out = argv[3];
    int n_frames = argc > 4 ? std::atoi(argv[4]) : -1;
    lmono_camera cam = { 1241, 376, 718.856, 718.856, 607.1928, 185.2157, 0, 0, 0, 0, 5, 0, 0 };   // kitti00_cam.yaml, kitti_map_config_00.yaml
    if (argc > 10) { cam.width = std::atoi(argv[5]); cam.height = std::atoi(argv[6]); cam.fx = std::atof(argv[7]); cam.fy = std::atof(argv[8]); cam.cx = std::atof(argv[9]); cam.cy = std::atof(argv[10]); }
    // camera-from-LiDAR extrinsic of the KITTI rig (camera x right, y down, z forward; p_l = rlc p_c + tlc)
    const double rlc[9] = { 0, 0, 1, -1, 0, 0, 0, -1, 0 }, tlc[3] = { 0.27, 0.0, -0.08 };
    std::vector<std::array<double, 8>> poses;
    {
        FILE *f = std::fopen(traj.c_str(), "r");
        if (!f) { std::fprintf(stderr, "cannot read %s\n", traj.c_str()); return 1; }
        std::array<double, 8> p;
        while (std::fscanf(f, "%lf %lf %lf %lf %lf %lf %lf %lf", &p[0], &p[1], &p[2], &p[3], &p[4], &p[5], &p[6], &p[7]) == 8) poses.push_back(p);
        std::fclose(f);
    }
    if (n_frames < 0 || n_frames > (int)poses.size()) n_frames = (int)poses.size();
    try {
        HipContext hip(0);
        MapBuilder map_builder(hip, cam, true, out);
        MappingLog recorder(out + "/mapping_recorder.txt");
        if (!recorder.ok()) { std::fprintf(stderr, "cannot write into %s\n", out.c_str()); return 1; }
        std::vector<uint8_t> image;
        for (int k = 0; k < n_frames; k++) {
            std::vector<float> xyzi;
            const long n = read_velodyne_bin(velodyne_path(seq, k), xyzi);
            char name[64];
            std::snprintf(name, sizeof name, "/image_bgr/%06d.bgr", k);
            if (n < 0 || !read_file(seq + name, image, (size_t)cam.width * cam.height * 3)) { std::fprintf(stderr, "frame %d unreadable\n", k); return 1; }
            const double *p = poses[(size_t)k].data();
            const auto t0 = std::chrono::steady_clock::now();
            const int m = map_builder.associateToMap(p + 4, p + 1, xyzi.data(), (int)n, rlc, tlc, image.data(), p[0]);
            recorder.write(p[0], std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            const std::string written = map_builder.processMapping();
            std::printf("FRAME %d points_in %ld points_out %d%s%s\n", k, n, m, written.empty() ? "" : " wrote ", written.c_str());
            if (k == n_frames - 1) {
                if (!dump(out + "/depth_last.u8", map_builder.depthMap()) || !dump(out + "/cloud_cam_last.bin", map_builder.rgbCloud(0))) return 1;
                if (written.empty() && !dump(out + "/cloud_world_last.bin", map_builder.rgbCloud(1))) return 1;
            }
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "map_build: %s\n", e.what());
        return 1;
    }
    return 0;
}
