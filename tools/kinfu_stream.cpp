// kinfu_stream -- BASELINE configs[2] driven from C++ with no Python anywhere: a TUM-layout directory -> TUMDataLoader ->
// frames resident in HBM -> per frame bilateral filter, integrate, ray cast + normals through tsdf_pipeline_step (the two-stream
// schedule of tsdf_amd/csrc/pipeline.hip) -> one JSON line with the time per step and a checksum of the last picture.
//
// The loop is the reference's src/Tools/kinfu.cpp:32-56 (TUMDataLoader::next, Camera::set_pose, TSDFVolume::integrate, frame after
// frame from ground-truth poses) with the filter and a ray cast added per frame, as BASELINE configs[2] asks, and with the frames
// uploaded once instead of one blocking host call per frame.  bench.py --tum-dir <same directory> runs the same frames through
// the same entry points from Python: same checksum, same ms per step (tools/compare_drivers.sh).
//
//   kinfu_stream -d <tum dir> [-n grid=512] [-p physical_mm=3000] [-k steps=20] [-w warmup=5] [--no-overlap]
//                [--no-cull-ahead] [--dump <dir>] [--track]
//   --track: BASELINE configs[4]'s loop instead -- the first frame at its ground-truth pose, every later one tracked against the model
//           (tsdf_tracker_filter / _align / _integrate: what src/Tools/tsdf_icp.cpp:115-198 does for one frame, composed with kinfu's
//           integrate; the pose is composed here with the Camera class); the first -k frames of the directory, one JSON line with the
//           time per frame and the distance of the last pose from its ground truth; --dump writes the tracked poses (poses.f32)
//   --dump: the last picture (vertices.f32, normals.f32), the final volume (distances.f32, weights.f32) and every frame's
//           pose (poses.f32, 16 floats each, column-major) as raw files, for tests/test_cpp_stream.py
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "Camera.hpp"
#include "DepthImage.hpp"
#include "TUMDataLoader.hpp"
#include "tsdf_amd.h"

static void ok(int rc, const char *what) {
    if (rc != TSDF_OK) {
        std::fprintf(stderr, "kinfu_stream: %s: %s\n", what, tsdf_last_error());
        std::exit(1);
    }
}

static void dump(const std::string &path, const void *p, size_t bytes) {
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)p, (std::streamsize)bytes);
    if (!f) {
        std::fprintf(stderr, "kinfu_stream: cannot write %s\n", path.c_str());
        std::exit(1);
    }
}

static tsdf_camera_matrices matrices_of(const Camera &cam) {
    tsdf_camera_matrices m;
    const Eigen::Matrix3f k = cam.k(), kinv = cam.kinv();
    std::memcpy(m.pose, cam.pose().data(), sizeof(m.pose));
    std::memcpy(m.inv_pose, cam.inverse_pose().data(), sizeof(m.inv_pose));
    std::memcpy(m.k, k.data(), sizeof(m.k));
    std::memcpy(m.kinv, kinv.data(), sizeof(m.kinv));
    return m;
}

int main(int argc, char **argv) {
    std::string dir, dump_dir;
    unsigned n = 512;
    float physical = 3000.0f;
    int K = 20, Wu = 5;
    bool overlap = true, cull_ahead = true, track = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto value = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "kinfu_stream: %s needs a value\n", a.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "-d") dir = value();
        else if (a == "-n") n = (unsigned)std::atoi(value());
        else if (a == "-p") physical = (float)std::atof(value());
        else if (a == "-k") K = std::atoi(value());
        else if (a == "-w") Wu = std::atoi(value());
        else if (a == "--no-overlap") overlap = false;
        else if (a == "--no-cull-ahead") cull_ahead = false;
        else if (a == "--dump") dump_dir = value();
        else if (a == "--track") track = true;
        else {
            std::fprintf(stderr, "usage: kinfu_stream -d <tum dir> [-n grid] [-p physical_mm] [-k steps] [-w warmup] [--no-overlap] [--no-cull-ahead] [--dump dir] [--track]\n");
            return 2;
        }
    }
    if (dir.empty() || K < 1 || Wu < 0 || n < 1) {
        std::fprintf(stderr, "kinfu_stream: -d <tum dir>, -k >= 1, -w >= 0, -n >= 1\n");
        return 2;
    }

    // ---- the stream: every frame of the directory, in millimetres, with its ground-truth pose (kinfu.cpp:32-51) ----------
    std::vector<std::vector<uint16_t>> frames;
    std::vector<tsdf_camera_matrices> cams;
    std::vector<Eigen::Matrix4f> truth;
    uint32_t W = 0, H = 0;
    try {
        TUMDataLoader loader(dir);
        std::unique_ptr<Camera> camera(Camera::default_depth_camera());
        Eigen::Matrix4f pose;
        while (DepthImage *di = loader.next(pose)) {
            std::unique_ptr<DepthImage> image(di);
            if (frames.empty()) {
                W = image->width();
                H = image->height();
            } else if (image->width() != W || image->height() != H) {
                throw std::invalid_argument("depth images of different sizes");
            }
            frames.emplace_back(image->data(), image->data() + (size_t)W * H);
            camera->set_pose(pose);
            cams.push_back(matrices_of(*camera));
            truth.push_back(pose);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "kinfu_stream: %s\n", e.what());
        return 1;
    }
    if (frames.empty()) {
        std::fprintf(stderr, "kinfu_stream: no frames in %s\n", dir.c_str());
        return 1;
    }
    const size_t F = frames.size(), n_pix = (size_t)W * H;

    // ---- everything resident in HBM before the clock starts ---------------------------------------------------------------
    uint16_t *depth_dev = nullptr;
    float *vert_dev = nullptr, *norm_dev = nullptr;
    ok(tsdf_device_alloc(F * n_pix * sizeof(uint16_t), (void **)&depth_dev), "frames");
    for (size_t i = 0; i < F; i++) ok(tsdf_device_upload(depth_dev + i * n_pix, frames[i].data(), n_pix * sizeof(uint16_t)), "frames");
    ok(tsdf_device_alloc(n_pix * 3 * sizeof(float), (void **)&vert_dev), "vertex map");
    ok(tsdf_device_alloc(n_pix * 3 * sizeof(float), (void **)&norm_dev), "normal map");
    tsdf_volume *vol = nullptr;
    tsdf_bilateral *bil = nullptr;
    tsdf_pipeline *pipe = nullptr;
    ok(tsdf_volume_create(n, n, n, physical, physical, physical, &vol), "volume");
    ok(tsdf_bilateral_create(30.0f, 4.5f, &bil), "bilateral filter");

    if (track) {
        // ---- the tracked loop: pose[i] = pose[i-1] * T, T from ICP of the filtered frame against the model rendered from pose[i-1] ----
        std::unique_ptr<Camera> camera(Camera::default_depth_camera());
        const Eigen::Matrix3f k = camera->k();
        tsdf_icp *icp = nullptr;
        tsdf_tracker *trk = nullptr;
        ok(tsdf_icp_create((int)W, (int)H, k(0, 2), k(1, 2), k(0, 0), k(1, 1), 0.10f, sinf(20.f * 3.14159254f / 180.f), &icp), "ICP");   // ICPOdometry.h:27
        ok(tsdf_tracker_create(vol, bil, icp, W, H, 20.0f, overlap ? TSDF_PIPELINE_OVERLAP : 0, &trk), "tracker");
        const size_t n_track = std::min(F, (size_t)K);
        std::vector<float> tracked;
        float error = 0.f, inliers = 0.f;
        camera->set_pose(truth[0]);
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        const size_t first_timed = n_track > 4 ? 4 : 1;     // (the first frames also allocate scratch and build the ray caster's flags)
        for (size_t i = 0; i < n_track; i++) {
            if (i == first_timed) {
                ok(tsdf_tracker_synchronize(trk), "synchronize");
                t0 = std::chrono::steady_clock::now();
            }
            ok(tsdf_tracker_filter(trk, depth_dev + i * n_pix), "filter");
            if (i > 0) {
                const tsdf_camera_matrices prev = matrices_of(*camera);
                double T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};     // column-major; current camera -> previous camera, metres
                ok(tsdf_tracker_align(trk, &prev, T, &error, &inliers), "align");
                // pose <- pose * T with T's translation in millimetres, in double, narrowed once (as tsdf_amd/tracking.py does)
                Eigen::Matrix4f next;
                const Eigen::Matrix4f pose = camera->pose();
                for (int r = 0; r < 4; r++)
                    for (int c = 0; c < 4; c++) {
                        double s = 0.0;
                        for (int j = 0; j < 4; j++) s += (double)pose(r, j) * (T[c * 4 + j] * ((c == 3 && j < 3) ? 1000.0 : 1.0));
                        next(r, c) = (float)s;
                    }
                camera->set_pose(next);
            }
            const tsdf_camera_matrices now = matrices_of(*camera);
            ok(tsdf_tracker_integrate(trk, &now), "integrate");
            tracked.insert(tracked.end(), now.pose, now.pose + 16);
        }
        ok(tsdf_tracker_synchronize(trk), "synchronize");
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const Eigen::Matrix4f last = camera->pose(), want = truth[n_track - 1];
        double dt2 = 0.0;
        for (int r = 0; r < 3; r++) dt2 += ((double)last(r, 3) - want(r, 3)) * ((double)last(r, 3) - want(r, 3));
        std::printf("{\"driver\": \"tools/kinfu_stream.cpp --track (C++, tsdf_tracker_*)\", \"grid\": %u, \"image\": [%u, %u], \"frames\": %zu, \"overlap\": %s, "
                    "\"ms_per_frame\": %.4f, \"last_pose_translation_error_mm\": %.4f, \"last_icp_inliers\": %.0f, \"last_icp_error\": %.6g}\n",
                    n, W, H, n_track, overlap ? "true" : "false", n_track > first_timed ? elapsed * 1e3 / (double)(n_track - first_timed) : 0.0, std::sqrt(dt2),
                    inliers, error);
        if (!dump_dir.empty()) {
            dump(dump_dir + "/poses.f32", tracked.data(), tracked.size() * sizeof(float));
            std::vector<float> a((size_t)n * n * n);
            ok(tsdf_volume_get_distance_data(vol, a.data()), "distances");
            dump(dump_dir + "/distances.f32", a.data(), a.size() * sizeof(float));
        }
        ok(tsdf_tracker_destroy(trk), "tracker");
        tsdf_icp_destroy(icp);
        ok(tsdf_bilateral_destroy(bil), "bilateral filter");
        ok(tsdf_volume_destroy(vol), "volume");
        (void)tsdf_device_free(depth_dev);
        (void)tsdf_device_free(vert_dev);
        (void)tsdf_device_free(norm_dev);
        return 0;
    }
    ok(tsdf_pipeline_create(vol, bil, W, H, overlap ? TSDF_PIPELINE_OVERLAP : 0, nullptr, &pipe), "pipeline");

    auto step = [&](int i) {
        const size_t f = (size_t)i % F, g = (size_t)(i + 1) % F;
        // (every step announces its successor, the last timed one too -- as bench.py does)
        ok(tsdf_pipeline_step(pipe, depth_dev + f * n_pix, &cams[f], vert_dev, norm_dev, depth_dev + g * n_pix, cull_ahead ? &cams[g] : nullptr), "step");
    };
    for (int i = 0; i < Wu; i++) step(i);
    ok(tsdf_pipeline_synchronize(pipe), "synchronize");
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = Wu; i < Wu + K; i++) step(i);
    ok(tsdf_pipeline_synchronize(pipe), "synchronize");
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // ---- the last picture: checksum = sum of the 32-bit patterns of every word, as signed integers (order independent, exact) --
    std::vector<float> V(n_pix * 3), N(n_pix * 3);
    ok(tsdf_device_download(V.data(), vert_dev, V.size() * sizeof(float)), "vertex map");
    ok(tsdf_device_download(N.data(), norm_dev, N.size() * sizeof(float)), "normal map");
    long long bits_v = 0, bits_n = 0, hits = 0;
    for (size_t i = 0; i < V.size(); i++) {
        int32_t w;
        std::memcpy(&w, &V[i], 4);
        bits_v += w;
        std::memcpy(&w, &N[i], 4);
        bits_n += w;
    }
    for (size_t i = 0; i < n_pix; i++) hits += !std::isnan(V[3 * i]);
    const double ms = elapsed * 1e3 / K, voxels = (double)n * n * n;
    std::printf("{\"driver\": \"tools/kinfu_stream.cpp (C++, tsdf_pipeline_step)\", \"grid\": %u, \"image\": [%u, %u], \"frames_in_directory\": %zu, "
                "\"steps\": %d, \"warmup\": %d, \"overlap\": %s, \"cull_ahead\": %s, \"ms_per_step\": %.4f, \"value\": %.3f, \"unit\": \"Mvoxels/s\", "
                "\"last_frame_vertex_bits\": %lld, \"last_frame_normal_bits\": %lld, \"last_frame_hits\": %lld}\n",
                n, W, H, F, K, Wu, overlap ? "true" : "false", cull_ahead ? "true" : "false", ms, voxels * K / elapsed / 1e6, bits_v, bits_n, hits);

    if (!dump_dir.empty()) {
        dump(dump_dir + "/vertices.f32", V.data(), V.size() * sizeof(float));
        dump(dump_dir + "/normals.f32", N.data(), N.size() * sizeof(float));
        std::vector<float> a((size_t)n * n * n);
        ok(tsdf_volume_get_distance_data(vol, a.data()), "distances");
        dump(dump_dir + "/distances.f32", a.data(), a.size() * sizeof(float));
        ok(tsdf_volume_get_weight_data(vol, a.data()), "weights");
        dump(dump_dir + "/weights.f32", a.data(), a.size() * sizeof(float));
        std::vector<float> poses;
        for (const tsdf_camera_matrices &m : cams) poses.insert(poses.end(), m.pose, m.pose + 16);
        dump(dump_dir + "/poses.f32", poses.data(), poses.size() * sizeof(float));
    }

    ok(tsdf_pipeline_destroy(pipe), "pipeline");
    ok(tsdf_bilateral_destroy(bil), "bilateral filter");
    ok(tsdf_volume_destroy(vol), "volume");
    (void)tsdf_device_free(depth_dev);
    (void)tsdf_device_free(vert_dev);
    (void)tsdf_device_free(norm_dev);
    return 0;
}
