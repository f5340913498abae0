// Host-side diagnostics of integrate's brick list, compiled only with -DTSDF_DIAGNOSTICS (`make DIAG=1`): none of this is in
// the product library.  Up to round 2 these paths lived inside launch_integrate (integrate.hip); every one of them synchronises
// the stream and copies the list to the host, so they have no place in the launcher.
//   TSDF_DEBUG_SORT = 1..8   the integrate brick list re-sorted on the host before the launch, to time integrate_kernel on other
//                            orders: 1 index order, 2 scattered, 3 position in the layer then layer, 4 x / z / y (what the cull
//                            kernel produces), 5 x / y / z, 6 x then scattered rows, 7 z / x / y, 8 even rows first
//   TSDF_DEBUG_SORT = 9 / 10 (with TSDF_DEBUG_BRICKS=3) the previous launch's longest bricks first, in 2 / 3 classes
//   TSDF_DEBUG_BRICKS = 1    bricks surviving the cull; 2 = + order statistics of the list; 3 = + per-brick clocks of the launch
#ifdef TSDF_DIAGNOSTICS
#include <algorithm>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#include "integrate_grid.hpp"

namespace tsdf {

static std::unordered_map<uint32_t, float> g_prev_brick_us;   // brick -> microseconds in the previous launch (TSDF_DEBUG_SORT=9/10)

void diag_sort_brick_list(tsdf_volume *v, const BrickGrid &bg, uint32_t *count, uint4 *boxes) {
    if (!v->nodes) {
        static const int sort_mode = [] { const char *e = getenv("TSDF_DEBUG_SORT"); return e ? atoi(e) : 0; }();
        // 9 / 10 (with TSDF_DEBUG_BRICKS=3): bricks that took long in the previous launch first, in 2 / 3 classes, each class column by column
        if (sort_mode >= 9 && !g_prev_brick_us.empty()) {
            (void)hipStreamSynchronize(v->stream);
            uint32_t n = 0;
            (void)hipMemcpy(&n, count, sizeof(n), hipMemcpyDeviceToHost);
            std::vector<uint32_t> l(n), idx(n), l2(n);
            std::vector<uint4> bx(n), bx2(n);
            (void)hipMemcpy(l.data(), v->brick_list, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
            (void)hipMemcpy(bx.data(), boxes, n * sizeof(uint4), hipMemcpyDeviceToHost);
            std::vector<float> known;
            for (uint32_t i = 0; i < n; i++) { idx[i] = i; auto it = g_prev_brick_us.find(l[i]); if (it != g_prev_brick_us.end()) known.push_back(it->second); }
            std::sort(known.begin(), known.end());
            const int classes = sort_mode == 9 ? 2 : 3;
            auto cls = [&](uint32_t i) -> uint64_t {
                auto it = g_prev_brick_us.find(l[i]);
                if (it == g_prev_brick_us.end() || known.empty()) return 0;   // unknown: with the expensive ones
                const size_t rank = std::lower_bound(known.begin(), known.end(), it->second) - known.begin();
                return (uint64_t)(classes - 1 - std::min<size_t>(classes - 1, rank * classes / known.size()));
            };
            std::vector<uint64_t> key(n);
            for (uint32_t i = 0; i < n; i++) key[i] = (cls(i) << 56) | ((uint64_t)(l[i] % bg.nx) << 32) | (l[i] / bg.nx);
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t c) { return key[a] < key[c]; });
            for (uint32_t i = 0; i < n; i++) { l2[i] = l[idx[i]]; bx2[i] = bx[idx[i]]; }
            (void)hipMemcpy(v->brick_list, l2.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice);
            (void)hipMemcpy(boxes, bx2.data(), n * sizeof(uint4), hipMemcpyHostToDevice);
        } else if (sort_mode && sort_mode < 9) {
            (void)hipStreamSynchronize(v->stream);
            uint32_t n = 0;
            (void)hipMemcpy(&n, count, sizeof(n), hipMemcpyDeviceToHost);
            std::vector<uint32_t> l(n), idx(n);
            std::vector<uint4> bx(n), bx2(n);
            (void)hipMemcpy(l.data(), v->brick_list, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
            (void)hipMemcpy(bx.data(), boxes, n * sizeof(uint4), hipMemcpyDeviceToHost);
            for (uint32_t i = 0; i < n; i++) idx[i] = i;
            const uint32_t layer = bg.nx * bg.ny;
            auto key = [&](uint32_t i) -> uint64_t {
                const uint32_t b = l[i];
                if (sort_mode == 1) return b;
                if (sort_mode == 2) return ((uint64_t)b * 2654435761u) & 0xffffffffu;
                if (sort_mode == 3) return ((uint64_t)(b % layer) << 8) | (b / layer);           // position in the layer, then the layer
                if (sort_mode == 4) return ((uint64_t)(b % bg.nx) << 32) | (b / bg.nx);             // x, then the row
                const uint32_t bxx = b % bg.nx, byy = (b / bg.nx) % bg.ny, bzz = b / layer;
                if (sort_mode == 5) return ((uint64_t)bxx << 32) | ((uint64_t)byy << 16) | bzz;      // x, y, then the layer
                if (sort_mode == 6) return ((uint64_t)bxx << 32) | (((uint64_t)(b / bg.nx) * 2654435761u) & 0xffffffffu);   // x, rows scattered
                if (sort_mode == 7) return ((uint64_t)bzz << 32) | ((uint64_t)bxx << 16) | byy;      // layer, x, y
                return ((uint64_t)(byy & 1u) << 48) | ((uint64_t)bxx << 32) | (b / bg.nx);          // even rows first, x, row
            };
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t c) { return key(a) < key(c); });
            std::vector<uint32_t> l2(n);
            for (uint32_t i = 0; i < n; i++) { l2[i] = l[idx[i]]; bx2[i] = bx[idx[i]]; }
            (void)hipMemcpy(v->brick_list, l2.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice);
            (void)hipMemcpy(boxes, bx2.data(), n * sizeof(uint4), hipMemcpyHostToDevice);
        }
    }
}

unsigned long long *diag_brick_log_alloc(tsdf_volume *v, size_t n_bricks) {
    static const int debug_clocks = [] { const char *e = getenv("TSDF_DEBUG_BRICKS"); return e && atoi(e) >= 3; }();
    unsigned long long *brick_log = nullptr;
    if (debug_clocks && !v->counting && !v->nodes) {
        (void)hipMalloc((void **)&brick_log, 2 * n_bricks * sizeof(unsigned long long));
        (void)hipMemset(brick_log, 0, 2 * n_bricks * sizeof(unsigned long long));
    }
    return brick_log;
}

void diag_brick_report(tsdf_volume *v, const BrickGrid &bg, size_t n_bricks, uint32_t *count, uint4 *boxes, unsigned long long *brick_log) {
    if (brick_log) {   // diagnostics (synchronises): the launch's bricks over time
        (void)hipStreamSynchronize(v->stream);
        uint32_t n = 0;
        (void)hipMemcpy(&n, count, sizeof(n), hipMemcpyDeviceToHost);
        std::vector<unsigned long long> log(2 * (size_t)n);
        (void)hipMemcpy(log.data(), brick_log, log.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        (void)hipFree(brick_log);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (uint32_t e = 0; e < n; e++) if (log[2 * e]) { t0 = std::min(t0, log[2 * e]); t1 = std::max(t1, log[2 * e + 1]); }
        {
            std::vector<uint32_t> lst(n);
            (void)hipMemcpy(lst.data(), v->brick_list, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
            g_prev_brick_us.clear();
            for (uint32_t e = 0; e < n; e++) g_prev_brick_us[lst[e]] = (float)((double)(log[2 * e + 1] - log[2 * e]) / 100.0);
        }
        size_t alive[16] = {};
        double dur = 0, dmax = 0, dmin = 1e18, last_start = 0;
        for (uint32_t e = 0; e < n; e++) {
            const double b0 = (double)(log[2 * e] - t0) / 100.0, e0 = (double)(log[2 * e + 1] - t0) / 100.0;
            dur += e0 - b0; dmax = std::max(dmax, e0 - b0); dmin = std::min(dmin, e0 - b0); last_start = std::max(last_start, b0);
            for (int q = 0; q < 16; q++) { const double tq = (q + 0.5) / 16.0 * (double)(t1 - t0) / 100.0; if (b0 <= tq && tq < e0) alive[q]++; }
        }
        fprintf(stderr, "tsdf: integrate_kernel %.1f us (100 MHz clock): %u bricks, %.1f us each (%.1f .. %.1f), last start %.1f; alive per sixteenth:", (t1 - t0) / 100.0, n,
                n ? dur / n : 0.0, dmin, dmax, last_start);
        for (int q = 0; q < 16; q++) fprintf(stderr, " %zu", alive[q]);
        fprintf(stderr, "\n");
        {   // the bricks that end last, and the longest ones: start, duration, pixel box
            std::vector<uint4> bx(n);
            (void)hipMemcpy(bx.data(), boxes, n * sizeof(uint4), hipMemcpyDeviceToHost);
            std::vector<uint32_t> idx(n);
            for (uint32_t e = 0; e < n; e++) idx[e] = e;
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t c) { return log[2 * a + 1] > log[2 * c + 1]; });
            fprintf(stderr, "tsdf:   last to end (entry: start + duration us, box):");
            for (uint32_t r = 0; r < std::min(n, 6u); r++) { const uint32_t e = idx[r]; fprintf(stderr, " %u: %.0f + %.0f, %ux%u;", e, (log[2 * e] - t0) / 100.0, (log[2 * e + 1] - log[2 * e]) / 100.0, bx[e].z, bx[e].w); }
            std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t c) { return log[2 * a + 1] - log[2 * a] > log[2 * c + 1] - log[2 * c]; });
            fprintf(stderr, "\ntsdf:   longest:");
            for (uint32_t r = 0; r < std::min(n, 6u); r++) { const uint32_t e = idx[r]; fprintf(stderr, " %u: %.0f + %.0f, %ux%u;", e, (log[2 * e] - t0) / 100.0, (log[2 * e + 1] - log[2 * e]) / 100.0, bx[e].z, bx[e].w); }
            size_t n_unstaged = 0; double d_unstaged = 0, d_staged = 0;
            for (uint32_t e = 0; e < n; e++) { const bool st = bx[e].z != 0 && ((bx[e].z + 1u) & ~1u) * bx[e].w <= (uint32_t)kTilePixels; const double d = (log[2 * e + 1] - log[2 * e]) / 100.0; if (st) d_staged += d; else { d_unstaged += d; n_unstaged++; } }
            fprintf(stderr, "\ntsdf:   %zu bricks without a tile: %.1f us each; the others %.1f us\n", n_unstaged, n_unstaged ? d_unstaged / n_unstaged : 0.0, n > n_unstaged ? d_staged / (n - n_unstaged) : 0.0);
        }
    }
    if (getenv("TSDF_DEBUG_BRICKS") && !v->nodes) {   // diagnostics: how many bricks survived the cull
        uint32_t n_active = 0;
        (void)hipMemcpy(&n_active, count, sizeof(n_active), hipMemcpyDeviceToHost);
        fprintf(stderr, "tsdf: %u of %zu bricks active (%.1f M voxels processed)\n", n_active, n_bricks, n_active * 4096.0 / 1e6);
        if (atoi(getenv("TSDF_DEBUG_BRICKS")) > 1) {   // the order of the list: its first entries, and how long its runs of consecutive bricks are
            std::vector<uint32_t> l(n_active);
            (void)hipMemcpy(l.data(), v->brick_list, n_active * sizeof(uint32_t), hipMemcpyDeviceToHost);
            fprintf(stderr, "tsdf: list starts");
            for (uint32_t i = 0; i < std::min(n_active, 40u); i++) fprintf(stderr, " %u", l[i]);
            size_t runs = 1, same_row = 0;
            for (uint32_t i = 1; i < n_active; i++) { runs += l[i] != l[i - 1] + 1; same_row += (l[i] / bg.nx == l[i - 1] / bg.nx); }
            fprintf(stderr, "\ntsdf: %zu runs of consecutive bricks (mean length %.1f), %zu neighbours in the same row\n", runs, (double)n_active / runs, same_row);
        }
    }
}

}  // namespace tsdf
#endif  // TSDF_DIAGNOSTICS
