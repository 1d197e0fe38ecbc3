// Host-only logic of libd2d.so that needs neither a device nor the HIP headers: candidate enumeration (the stand-in for
// differt-core's Rust graph iterator, differt2d/scene.py:153-175), parameter validation, lax.integer_pow, and the
// buffer-size arithmetic of the sweep launches (LDS tables, the contribution lists of the patches cut in four and their
// 4 GiB guard).  d2d.hip uses these functions as they are; tests/native/d2d_host_san.cpp compiles the same header with
// g++ -fsanitize=address,undefined and tests/test_host_sanitizers.py drives it (sanitizers run on the CPU build only).
#pragma once
#include <stdint.h>

#include <cmath>
#include <cstddef>
#include <string>
#include <vector>

#include "../../include/d2d.h"

namespace d2d_host {

// lax.integer_pow lowering (square and multiply), fp32
inline float integer_pow(float x, int n) {
    if (n == 0) return 1.0f;
    float acc = 0.0f;
    bool have = false;
    while (n > 0) {
        if (n & 1) {
            acc = have ? acc * x : x;
            have = true;
        }
        n >>= 1;
        if (n > 0) x = x * x;
    }
    return acc;
}

// 64-bit content hash of an fp32 array (bit patterns: -0.0 != 0.0, every NaN payload its own value), four independent
// multiply-xor lanes over 8-byte words so that the loop runs at memory speed.  d2d_set_grid uses it to recognise a grid it
// already holds (a caller of the reference's API passes X, Y with every call: scene.py:1803-1826).
inline uint64_t hash_floats(const float* p, size_t n, uint64_t seed) {
    const uint64_t K0 = 0x9E3779B97F4A7C15ull, K1 = 0xC2B2AE3D27D4EB4Full, K2 = 0x165667B19E3779F9ull, K3 = 0xD6E8FEB86659FD93ull;
    uint64_t h0 = seed ^ K0, h1 = seed ^ K1, h2 = seed ^ K2, h3 = seed ^ K3;
    size_t i = 0;
    auto word = [&](size_t at) {
        uint64_t w;
        __builtin_memcpy(&w, p + at, 8);
        return w;
    };
    for (; i + 8 <= n; i += 8) {
        h0 = (h0 ^ word(i)) * K1;
        h1 = (h1 ^ word(i + 2)) * K2;
        h2 = (h2 ^ word(i + 4)) * K3;
        h3 = (h3 ^ word(i + 6)) * K0;
        h0 ^= h0 >> 29;
        h1 ^= h1 >> 31;
        h2 ^= h2 >> 30;
        h3 ^= h3 >> 28;
    }
    for (; i < n; ++i) {
        uint32_t w;
        __builtin_memcpy(&w, p + i, 4);
        h0 = (h0 ^ (uint64_t)w) * K2;
        h0 ^= h0 >> 31;
    }
    uint64_t h = h0 ^ (h1 * K3) ^ (h2 * K0) ^ (h3 * K1) ^ ((uint64_t)n * K2);
    h ^= h >> 33;
    h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 33;
    h *= 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 33;
    return h;
}

// candidates of order k over nc allowed objects: nc (nc - 1)^(k-1), 1 for k = 0; saturates at INT64_MAX
inline int64_t count_order(int64_t nc, int k) {
    if (k == 0) return 1;
    if (nc <= 0) return 0;
    int64_t c = nc;
    for (int i = 1; i < k; ++i) {
        if (nc - 1 != 0 && c > INT64_MAX / (nc - 1)) return INT64_MAX;
        c *= (nc - 1);
    }
    return c;
}

// status (D2D_OK or negative) + message
inline int count_candidates(int32_t n_objects, const uint8_t* allowed, int32_t min_order, int32_t max_order, int64_t* count,
                            std::string& err) {
    if (!count) return err = "count is NULL", D2D_ERR_INVALID;
    if (n_objects < 0 || min_order < 0) return err = "negative argument", D2D_ERR_INVALID;
    int64_t nc = 0;
    for (int j = 0; j < n_objects; ++j) nc += (!allowed || allowed[j]) ? 1 : 0;
    int64_t total = 0;
    for (int k = min_order; k <= max_order; ++k) {
        const int64_t ck = count_order(nc, k);
        if (ck > INT64_MAX - total) return err = "the number of candidates overflows int64", D2D_ERR_INVALID;
        total += ck;
    }
    *count = total;
    return D2D_OK;
}

// For each order k ascending, every tuple of k allowed object indices with no two equal neighbours, lexicographic
// (recorded order: docs/source/notebooks/cost20120_helsinki_model.ipynb cell 20).  cand: [count][D2D_MAX_ORDER], -1 padded.
inline int enumerate_candidates(int32_t n_objects, const uint8_t* allowed, int32_t min_order, int32_t max_order, int32_t* cand,
                                int32_t* order, int64_t capacity, std::string& err) {
    if (min_order < 0 || max_order > D2D_MAX_ORDER) return err = "orders must lie in [0, " + std::to_string(D2D_MAX_ORDER) + "]", D2D_ERR_INVALID;
    int64_t total = 0;
    int rc = count_candidates(n_objects, allowed, min_order, max_order, &total, err);
    if (rc) return rc;
    if (capacity < total) return err = "capacity " + std::to_string(capacity) + " < " + std::to_string(total) + " candidates", D2D_ERR_INVALID;
    std::vector<int> cw;
    for (int j = 0; j < n_objects; ++j)
        if (!allowed || allowed[j]) cw.push_back(j);
    const int nc = (int)cw.size();
    int64_t at = 0;
    for (int k = min_order; k <= max_order; ++k) {
        if (k == 0) {
            if (cand) for (int i = 0; i < D2D_MAX_ORDER; ++i) cand[at * D2D_MAX_ORDER + i] = -1;
            if (order) order[at] = 0;
            ++at;
            continue;
        }
        // odometer over positions in the compact list: lexicographic, no equal neighbours
        int pos[D2D_MAX_ORDER];
        int depth = 0;
        pos[0] = -1;
        while (depth >= 0) {
            int p = pos[depth] + 1;
            if (depth > 0 && p == pos[depth - 1]) ++p;
            if (p >= nc) {
                --depth;
                continue;
            }
            pos[depth] = p;
            if (depth == k - 1) {
                if (cand)
                    for (int i = 0; i < D2D_MAX_ORDER; ++i) cand[at * D2D_MAX_ORDER + i] = (i < k) ? cw[pos[i]] : -1;
                if (order) order[at] = k;
                ++at;
            } else {
                ++depth;
                pos[depth] = -1;
            }
        }
    }
    return D2D_OK;
}

inline int check_params(const d2d_params* p, std::string& err) {
    if (!p) return err = "params is NULL", D2D_ERR_INVALID;
    if (p->min_order < 0 || p->max_order > D2D_MAX_ORDER)
        return err = "orders must lie in [0, " + std::to_string(D2D_MAX_ORDER) + "], got [" + std::to_string(p->min_order) + ", " +
                     std::to_string(p->max_order) + "]", D2D_ERR_INVALID;
    if (p->approx && !(p->alpha > 0.0f)) return err = "alpha must be > 0 in approx mode", D2D_ERR_INVALID;
    if (p->approx && p->act != D2D_ACT_HARD_SIGMOID && p->act != D2D_ACT_SIGMOID)
        return err = "activation " + std::to_string(p->act) + " is not one of the native activations", D2D_ERR_UNSUPPORTED;
    if (p->fun_id < 0 || p->fun_id > D2D_FUN_CUSTOM) return err = "fun_id " + std::to_string(p->fun_id) + " is not a native path function", D2D_ERR_UNSUPPORTED;
    if (p->out_mode != D2D_OUT_OVERWRITE && p->out_mode != D2D_OUT_ADD) return err = "bad out_mode " + std::to_string(p->out_mode), D2D_ERR_INVALID;
    if (p->grid_role != D2D_GRID_RX && p->grid_role != D2D_GRID_TX) return err = "bad grid_role " + std::to_string(p->grid_role), D2D_ERR_INVALID;
    if (!(p->seg_tol >= 0.0f)) return err = "seg_tol must be >= 0", D2D_ERR_INVALID;
    return D2D_OK;
}

// ---- buffer sizes of the forward sweep ----------------------------------------------------------------------------
constexpr size_t LDS_LIMIT = 64 * 1024;  // dynamic LDS the launches that have a choice ask for (of the CU's 160 KB: several workgroups stay resident)
constexpr size_t LDS_MAX = 156 * 1024;    // ... and what a launch without a choice may take: one workgroup per CU (gfx950 grants a workgroup the
                                          // CU's whole 160 KB without an attribute, scripts/probes/lds_limit_probe.hip; 4 KB left for static LDS)
constexpr size_t F4 = 16;                // sizeof(float4)

// one-wave-per-patch kernels: [2N] refl + [N] flt + [N] adjoint table (float4 each) + 1 spare + one 512-byte culling queue
inline size_t tab_lds_bytes(int n_objects) { return (size_t)(4 * (size_t)n_objects + 1) * F4 + 512; }

struct SplitLds {
    size_t base;   // byte offset of the culling queues (16-byte aligned)
    size_t total;  // dynamic LDS of power_fwd_split_kernel
};
// shared-patch kernel (W waves): tables, (W - 1) contribution lists of list_len x 64 floats, their bookkeeping, W queues
inline SplitLds split_lds_bytes(int n_objects, int W, int list_len) {
    SplitLds s;
    s.base = ((tab_lds_bytes(n_objects) - 512 + (size_t)(W - 1) * (size_t)list_len * 64 * sizeof(float) +
               (size_t)((W - 1) * 65 + W + 1) * sizeof(int)) + 15) & ~(size_t)15;
    s.total = s.base + (size_t)W * 512;
    return s;
}

// candidate-sharing kernel (W waves, power_fwd_coop_kernel): tables + the rounds' slots [W][C][64]
inline size_t coop_lds_bytes(int n_objects, int W, int C) {
    return (size_t)(3 * n_objects + 1) * 16 + (size_t)W * (size_t)C * 64 * sizeof(float);
}

struct HeavyPlan {
    long long H = 0;           // patches cut in `parts` (0: none)
    long long cap = 0;         // list entries per lane and part
    long long list_floats = 0; // heavy_list elements
    long long cnt_ints = 0;    // heavy_cnt elements
};
// The dearest `heavy_split` patches of a launch of `tiles` patches (at most a quarter of them) are cut in `parts`; a
// part's list holds as many entries as it has candidates at most (orders 0..2 over Nc allowed walls), and the lists of
// all parts together must stay below 4 GiB (fewer patches are cut to fit; none if one does not fit).
inline HeavyPlan heavy_plan(long long tiles, long long Nc, long long heavy_split, long long parts) {
    HeavyPlan hp;
    if (tiles <= 0 || Nc < 2 || heavy_split <= 0 || parts <= 0) return hp;
    long long H = heavy_split < tiles / 4 ? heavy_split : tiles / 4;
    if (H <= 0) return hp;
    if (Nc > (1ll << 30)) return hp;
    // what one part can push: its range of first walls (enumerating build), or -- region lists, parts by rank, part 0 a
    // smaller share because it also sweeps orders 0 and 1 -- up to 1 / (parts - 1) of all order-2 candidates
    const long long by_walls = ((Nc + parts - 1) / parts + 2) * Nc + Nc + 2;
    const long long by_rank = (Nc * (Nc - 1) + (parts > 1 ? parts - 2 : 0)) / (parts > 1 ? parts - 1 : 1) + Nc + 2;
    const long long cap = by_walls > by_rank ? by_walls : by_rank;
    // H * parts * cap * 64 * 4 bytes <= 4 GiB, evaluated without overflow
    const long long limit = (4ll << 30) / (64 * (long long)sizeof(float));
    if (cap <= 0 || cap > limit / parts) return hp;
    if (H > limit / (parts * cap)) H = limit / (parts * cap);  // as many as fit
    if (H <= 0) return hp;
    hp.H = H;
    hp.cap = cap;
    hp.list_floats = H * parts * cap * 64;
    hp.cnt_ints = H * parts * 64 + H * parts;
    return hp;
}

// ---- region candidate lists (region_list_kernel / region_refine_kernel) ----------------------------------------------
struct RegionLevelPlan {
    int R = 0, S = 0;  // region = R x R patches; S lists (slices of first walls) per region
    int regions_x = 0, regions_y = 0;
    long long regions = 0, slots = 0;  // slots = regions * S = lists per order
};
struct RegionPlan {
    bool on = false;
    RegionLevelPlan top, leaf;  // top: listed by enumeration, S slices; leaf: one list per region, refined from top
    long long n_static = 0;     // chunks that are some list's first chunk
    long long max_chunks = 0;   // pool chunks of `chunk` entries
    int k_lo = 2;               // lists exist for the orders [k_lo, max_order]
};
// Lists exist for the orders k in [max(2, min_order), max_order].  R_top is rounded down to a multiple of R_leaf (at
// least R_leaf).  The pool holds budget_bytes worth of 8-byte entries in chunks of `chunk`; every list owns one chunk.
inline RegionPlan region_plan(int tiles_x, int tiles_y, long long Nc, int min_order, int max_order, int R_leaf, int R_top, int S_req,
                              long long budget_bytes, int chunk) {
    RegionPlan rp;
    if (tiles_x <= 0 || tiles_y <= 0 || Nc < 2 || max_order < 2 || R_leaf <= 0 || chunk <= 0 || budget_bytes <= 0) return rp;
    int S = S_req > 0 ? S_req : (int)((Nc + 3) / 4);
    if (S > 1024) S = 1024;
    if (S <= 0) return rp;
    if (R_top < R_leaf) R_top = R_leaf;
    R_top = (R_top / R_leaf) * R_leaf;
    auto level = [&](int R, int Sl) {
        RegionLevelPlan l;
        l.R = R;
        l.S = Sl;
        l.regions_x = (int)(((long long)tiles_x + R - 1) / R);
        l.regions_y = (int)(((long long)tiles_y + R - 1) / R);
        l.regions = (long long)l.regions_x * l.regions_y;  // < 2^62
        l.slots = l.regions > 0x3fffffffLL ? -1 : l.regions * Sl;  // (refused below; S <= 1024: no overflow)
        return l;
    };
    rp.leaf = level(R_leaf, 1);
    rp.top = level(R_top, S);
    if (rp.leaf.slots <= 0 || rp.leaf.slots > 0x3fffffffLL || rp.top.slots <= 0 || rp.top.slots > 0x3fffffffLL) return rp;
    rp.k_lo = min_order > 2 ? min_order : 2;
    if (rp.k_lo > max_order) return rp;
    rp.n_static = (rp.leaf.slots + rp.top.slots) * (max_order - rp.k_lo + 1);
    rp.max_chunks = budget_bytes / ((long long)sizeof(unsigned long long) * chunk);
    if (rp.max_chunks > 0x7fffffffLL) rp.max_chunks = 0x7fffffffLL;
    if (rp.n_static > 0x7fffffffLL || rp.max_chunks < 2 * rp.n_static) return rp;  // not worth having
    rp.on = true;
    return rp;
}

}  // namespace d2d_host
