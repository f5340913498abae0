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
//                [--no-cull-ahead] [--dump <dir>] [--track] [--ranks P] [--share-gpu] [--planes Z] [--validate-merge]
//   --planes Z: a grid of n x n x Z voxels over the same physical cube (flat voxels along z): many planes to shard at a small cost
//   --validate-merge (with --ranks): SURVEY.md 8e mode B after the timed steps -- every rank all-gathers the distance slabs, casts the
//           whole volume the single-volume way and compares its bits with the merged picture (tsdf_slab_validate_merge)
//   --ranks P: the volume in P Z-slabs, one PROCESS per slab (fork, before anything touches the GPU), each on its own GPU: slab
//           integrate + slab ray cast, the frame's all-gather of 8-byte hit records (tsdf_slab_exchange_*: RCCL on the step's
//           stream, the 128-byte id from rank 0 through shared memory) and the min-k merge on every rank, all through
//           tsdf_pipeline_step -- SURVEY.md 8e with no Python and no torch anywhere.  Rank 0 prints the line; the merged picture's
//           checksum must be the single-volume run's.  With fewer GPUs than ranks (or --share-gpu) every rank uses GPU 0 and the
//           records travel through host shared memory (a tsdf_exchange_fn): the N > 1 code path on a one-GPU box, timings meaningless.
//   --track: BASELINE configs[4]'s loop instead -- the first frame at its ground-truth pose, every later one tracked against the model
//           (tsdf_tracker_filter / _align / _integrate: what src/Tools/tsdf_icp.cpp:115-198 does for one frame, composed with kinfu's
//           integrate; the pose is composed here with the Camera class); the first -k frames of the directory, one JSON line with the
//           time per frame and the distance of the last pose from its ground truth; --dump writes the tracked poses (poses.f32)
//   --dump: the last picture (vertices.f32, normals.f32), the final volume (distances.f32, weights.f32) and every frame's
//           pose (poses.f32, 16 floats each, column-major) as raw files, for tests/test_cpp_stream.py
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

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

// ---- --ranks P: what the processes share (one anonymous shared mapping made before the fork) -------------------------------
constexpr int kMaxRanks = 64;
struct Shared {
    pthread_barrier_t barrier;
    uint8_t id[TSDF_EXCHANGE_ID_BYTES];
    double elapsed[kMaxRanks];
    long long bits_v[kMaxRanks], bits_n[kMaxRanks];
    unsigned long long merge_diff[kMaxRanks];
    int use_rccl;
};
struct ShmGather {   // user data of the host-staged all-gather (one GPU shared by every rank)
    Shared *shared;
    tsdf_hit_record *records;   // world x n_pixels, in the shared mapping
    int rank, world;
};
// tsdf_exchange_fn: rank r's records to slot r of the shared mapping, everybody's back to the device.  Blocking; correctness only.
static int shm_all_gather(void *user, const tsdf_hit_record *device_mine, tsdf_hit_record *device_all, uint32_t n_pixels, void *hip_stream) {
    ShmGather *g = (ShmGather *)user;
    if (tsdf_stream_synchronize(hip_stream) != TSDF_OK) return TSDF_ERR_DEVICE;
    if (tsdf_device_download(g->records + (size_t)g->rank * n_pixels, device_mine, (size_t)n_pixels * sizeof(tsdf_hit_record)) != TSDF_OK) return TSDF_ERR_DEVICE;
    pthread_barrier_wait(&g->shared->barrier);
    if (tsdf_device_upload(device_all, g->records, (size_t)n_pixels * g->world * sizeof(tsdf_hit_record)) != TSDF_OK) return TSDF_ERR_DEVICE;
    pthread_barrier_wait(&g->shared->barrier);   // (nobody overwrites its slot before everybody has read it)
    return TSDF_OK;
}

int main(int argc, char **argv) {
    std::string dir, dump_dir;
    int ranks = 1;
    bool share_gpu = false;
    unsigned n = 512;
    float physical = 3000.0f;
    int K = 20, Wu = 5;
    bool overlap = true, cull_ahead = true, track = false, validate_merge = false;
    unsigned planes = 0;
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
        else if (a == "--ranks") ranks = std::atoi(value());
        else if (a == "--share-gpu") share_gpu = true;
        else if (a == "--planes") planes = (unsigned)std::atoi(value());
        else if (a == "--validate-merge") validate_merge = true;
        else {
            std::fprintf(stderr, "usage: kinfu_stream -d <tum dir> [-n grid] [-p physical_mm] [-k steps] [-w warmup] [--no-overlap] [--no-cull-ahead] [--dump dir] [--track] [--ranks P] [--share-gpu] [--planes Z] [--validate-merge]\n");
            return 2;
        }
    }
    if (planes == 0) planes = n;
    if (dir.empty() || K < 1 || Wu < 0 || n < 1 || ranks < 1 || ranks > kMaxRanks || (unsigned)ranks > planes || (track && ranks > 1)) {
        std::fprintf(stderr, "kinfu_stream: -d <tum dir>, -k >= 1, -w >= 0, -n >= 1, 1 <= --ranks <= min(%d, grid), --track is single-volume\n", kMaxRanks);
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

    // ---- --ranks P: one process per Z-slab, forked here -- nothing has touched the GPU yet, every child starts its own HIP runtime
    int rank = 0;
    Shared *shared = nullptr;
    tsdf_hit_record *shared_records = nullptr;
    if (ranks > 1) {
        const size_t bytes = sizeof(Shared) + n_pix * (size_t)ranks * sizeof(tsdf_hit_record);
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) {
            std::perror("kinfu_stream: mmap");
            return 1;
        }
        shared = new (m) Shared();
        shared_records = (tsdf_hit_record *)((char *)m + sizeof(Shared));
        pthread_barrierattr_t attr;
        pthread_barrierattr_init(&attr);
        pthread_barrierattr_setpshared(&attr, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&shared->barrier, &attr, (unsigned)ranks);
        std::fflush(stdout);
        std::fflush(stderr);
        std::vector<pid_t> kids;
        bool child = false;
        for (int r = 0; r < ranks && !child; r++) {
            const pid_t pid = fork();
            if (pid < 0) {
                std::perror("kinfu_stream: fork");
                for (pid_t k_ : kids) kill(k_, SIGKILL);
                return 1;
            }
            if (pid == 0) {
                rank = r;
                child = true;
            } else {
                kids.push_back(pid);
            }
        }
        if (!child) {   // the launcher: wait for the ranks; one failing takes the others (blocked in a barrier) with it
            int failed = 0;
            for (size_t left = kids.size(); left > 0; left--) {
                int status = 0;
                const pid_t done = wait(&status);
                if (done < 0) break;
                if (!(WIFEXITED(status) && WEXITSTATUS(status) == 0) && !failed) {
                    failed = 1;
                    for (pid_t k_ : kids) if (k_ != done) kill(k_, SIGKILL);
                }
            }
            return failed;
        }
    }
    int n_devices = 0;
    ok(tsdf_device_count(&n_devices), "device count");
    const bool use_rccl = ranks > 1 && !share_gpu && ranks <= n_devices;
    if (ranks > 1) ok(tsdf_set_device(use_rccl ? rank : 0), "set device");

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
    tsdf_slab_exchange *exch = nullptr;
    ShmGather shm_gather = {shared, shared_records, rank, ranks};
    const uint32_t z_begin = (uint32_t)((uint64_t)planes * rank / ranks), z_end = (uint32_t)((uint64_t)planes * (rank + 1) / ranks);   // equal plane counts
    if (ranks == 1) {
        ok(tsdf_volume_create(n, n, planes, physical, physical, physical, &vol), "volume");
    } else {
        ok(tsdf_volume_create_slab(n, n, planes, physical, physical, physical, z_begin, z_end, &vol), "slab volume");
        if (use_rccl) {
            if (rank == 0) ok(tsdf_slab_exchange_unique_id(shared->id, nullptr), "RCCL unique id");
            pthread_barrier_wait(&shared->barrier);
            ok(tsdf_slab_exchange_create(rank, ranks, shared->id, nullptr, &exch), "slab exchange (RCCL)");
        } else {
            ok(tsdf_slab_exchange_create_callback(rank, ranks, shm_all_gather, &shm_gather, &exch), "slab exchange (shared memory)");
        }
    }
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
            std::vector<float> a((size_t)n * n * planes);
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
    ok(tsdf_pipeline_create(vol, bil, W, H, overlap ? TSDF_PIPELINE_OVERLAP : 0, exch, &pipe), "pipeline");

    auto step = [&](int i) {
        const size_t f = (size_t)i % F, g = (size_t)(i + 1) % F;
        // (every step announces its successor, the last timed one too -- as bench.py does)
        ok(tsdf_pipeline_step(pipe, depth_dev + f * n_pix, &cams[f], vert_dev, norm_dev, depth_dev + g * n_pix, cull_ahead ? &cams[g] : nullptr), "step");
    };
    for (int i = 0; i < Wu; i++) step(i);
    ok(tsdf_pipeline_synchronize(pipe), "synchronize");
    if (shared) pthread_barrier_wait(&shared->barrier);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = Wu; i < Wu + K; i++) step(i);
    ok(tsdf_pipeline_synchronize(pipe), "synchronize");
    if (shared) pthread_barrier_wait(&shared->barrier);
    double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

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
    bool ranks_agree = true;
    if (shared) {   // the step is as long as its slowest rank; every rank holds the merged picture: the checksums must agree
        shared->elapsed[rank] = elapsed;
        shared->bits_v[rank] = bits_v;
        shared->bits_n[rank] = bits_n;
        pthread_barrier_wait(&shared->barrier);
        for (int r = 0; r < ranks; r++) {
            elapsed = std::max(elapsed, shared->elapsed[r]);
            ranks_agree = ranks_agree && shared->bits_v[r] == bits_v && shared->bits_n[r] == bits_n;
        }
    }
    // mode B: the merged picture against the ordinary cast of the all-gathered distance slabs, on every rank (collective)
    unsigned long long merge_diff = 0;
    int ranks_seen = ranks;
    if (shared) {
        ok(tsdf_slab_exchange_ranks_seen(exch, &ranks_seen), "ranks seen");
        if (validate_merge) {
            const tsdf_camera_matrices &last = cams[(size_t)(Wu + K - 1) % F];
            uint64_t d = 0;
            ok(tsdf_slab_validate_merge(vol, exch, W, H, last.pose, last.kinv, vert_dev, norm_dev, &d), "validate merge (mode B)");
            shared->merge_diff[rank] = d;
            pthread_barrier_wait(&shared->barrier);
            for (int r = 0; r < ranks; r++) merge_diff = std::max(merge_diff, shared->merge_diff[r]);
            ranks_agree = ranks_agree && merge_diff == 0;
        }
    }
    const double ms = elapsed * 1e3 / K, voxels = (double)n * n * planes;
    if (rank == 0) {
        if (ranks > 1)
            std::printf("{\"driver\": \"tools/kinfu_stream.cpp --ranks (C++, one process per Z-slab, tsdf_pipeline_step + tsdf_slab_exchange)\", \"ranks\": %d, "
                        "\"exchange\": \"%s\", \"ranks_seen\": %d, \"slab_planes\": %u, \"ranks_hold_the_same_picture\": %s, \"merge_validated_mode_b\": %s, "
                        "\"mode_b_differing_words\": %llu, ",
                        ranks, use_rccl ? "ncclAllGather on the step's stream (librccl, id through shared memory)" : "host shared memory, every rank on GPU 0 (timings meaningless)",
                        ranks_seen, z_end - z_begin, ranks_agree ? "true" : "false", validate_merge ? (merge_diff == 0 ? "true" : "false") : "null", merge_diff);
        else
            std::printf("{\"driver\": \"tools/kinfu_stream.cpp (C++, tsdf_pipeline_step)\", ");
        std::printf("\"grid\": %u, \"planes\": %u, \"image\": [%u, %u], \"frames_in_directory\": %zu, "
                    "\"steps\": %d, \"warmup\": %d, \"overlap\": %s, \"cull_ahead\": %s, \"ms_per_step\": %.4f, \"value\": %.3f, \"unit\": \"Mvoxels/s\", "
                    "\"last_frame_vertex_bits\": %lld, \"last_frame_normal_bits\": %lld, \"last_frame_hits\": %lld}\n",
                    n, planes, W, H, F, K, Wu, overlap ? "true" : "false", cull_ahead ? "true" : "false", ms, voxels * K / elapsed / 1e6, bits_v, bits_n, hits);
    }

    if (!dump_dir.empty() && ranks > 1) {   // the merged picture (rank 0) and every rank's slab of the volume (its own planes, without the halo)
        if (rank == 0) {
            dump(dump_dir + "/vertices.f32", V.data(), V.size() * sizeof(float));
            dump(dump_dir + "/normals.f32", N.data(), N.size() * sizeof(float));
        }
        tsdf_volume_info info;
        ok(tsdf_volume_get_info(vol, &info), "volume info");
        std::vector<float> a((size_t)n * n * (info.z_store_end - info.z_store_begin));
        ok(tsdf_volume_get_distance_data(vol, a.data()), "distances");
        dump(dump_dir + "/distances.rank" + std::to_string(rank) + ".f32", a.data(), (size_t)n * n * (z_end - z_begin) * sizeof(float));
    } else if (!dump_dir.empty()) {
        dump(dump_dir + "/vertices.f32", V.data(), V.size() * sizeof(float));
        dump(dump_dir + "/normals.f32", N.data(), N.size() * sizeof(float));
        std::vector<float> a((size_t)n * n * planes);
        ok(tsdf_volume_get_distance_data(vol, a.data()), "distances");
        dump(dump_dir + "/distances.f32", a.data(), a.size() * sizeof(float));
        ok(tsdf_volume_get_weight_data(vol, a.data()), "weights");
        dump(dump_dir + "/weights.f32", a.data(), a.size() * sizeof(float));
        std::vector<float> poses;
        for (const tsdf_camera_matrices &m : cams) poses.insert(poses.end(), m.pose, m.pose + 16);
        dump(dump_dir + "/poses.f32", poses.data(), poses.size() * sizeof(float));
    }

    ok(tsdf_pipeline_destroy(pipe), "pipeline");
    if (exch) ok(tsdf_slab_exchange_destroy(exch), "slab exchange");
    ok(tsdf_bilateral_destroy(bil), "bilateral filter");
    ok(tsdf_volume_destroy(vol), "volume");
    (void)tsdf_device_free(depth_dev);
    (void)tsdf_device_free(vert_dev);
    (void)tsdf_device_free(norm_dev);
    std::fflush(stdout);
    return (shared && !ranks_agree) ? 1 : 0;
}
