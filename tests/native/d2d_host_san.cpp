// C entry points over differt2d_amd/csrc/d2d_host.hpp -- the host-only logic of libd2d.so (candidate enumeration,
// parameter validation, launch buffer sizes) -- for the CPU sanitizer build:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared -fPIC
// (tests/test_host_sanitizers.py).  The product compiles the very same header into libd2d.so with hipcc.
#include "../../differt2d_amd/csrc/d2d_host.hpp"

#include <cstring>

extern "C" {

int san_count(int32_t n, const uint8_t* allowed, int32_t lo, int32_t hi, int64_t* count) {
    std::string err;
    return d2d_host::count_candidates(n, allowed, lo, hi, count, err);
}

int san_enumerate(int32_t n, const uint8_t* allowed, int32_t lo, int32_t hi, int32_t* cand, int32_t* order, int64_t capacity) {
    std::string err;
    return d2d_host::enumerate_candidates(n, allowed, lo, hi, cand, order, capacity, err);
}

int san_check_params(const d2d_params* p, char* msg, int cap) {
    std::string err;
    const int rc = d2d_host::check_params(p, err);
    if (msg && cap > 0) {
        std::strncpy(msg, err.c_str(), (size_t)cap - 1);
        msg[cap - 1] = 0;
    }
    return rc;
}

float san_integer_pow(float x, int n) { return d2d_host::integer_pow(x, n); }

uint64_t san_hash_floats(const float* p, uint64_t n, uint64_t seed) { return d2d_host::hash_floats(p, (size_t)n, seed); }

void san_lds(int n_objects, int W, int list_len, uint64_t* tab, uint64_t* split_base, uint64_t* split_total) {
    *tab = d2d_host::tab_lds_bytes(n_objects);
    const d2d_host::SplitLds s = d2d_host::split_lds_bytes(n_objects, W, list_len);
    *split_base = s.base;
    *split_total = s.total;
}

void san_heavy_plan(long long tiles, long long Nc, long long heavy_split, long long parts, long long* out4) {
    const d2d_host::HeavyPlan hp = d2d_host::heavy_plan(tiles, Nc, heavy_split, parts);
    out4[0] = hp.H;
    out4[1] = hp.cap;
    out4[2] = hp.list_floats;
    out4[3] = hp.cnt_ints;
}

// out: on, top.R, top.S, top.regions, top.slots, leaf.R, leaf.S, leaf.regions, leaf.slots, n_static, max_chunks, k_lo
void san_region_plan(int tiles_x, int tiles_y, long long Nc, int min_order, int max_order, int R_leaf, int R_top, int S_req,
                     long long budget_bytes, int chunk, long long* out12) {
    const d2d_host::RegionPlan rp = d2d_host::region_plan(tiles_x, tiles_y, Nc, min_order, max_order, R_leaf, R_top, S_req, budget_bytes, chunk);
    out12[0] = rp.on ? 1 : 0;
    out12[1] = rp.top.R; out12[2] = rp.top.S; out12[3] = rp.top.regions; out12[4] = rp.top.slots;
    out12[5] = rp.leaf.R; out12[6] = rp.leaf.S; out12[7] = rp.leaf.regions; out12[8] = rp.leaf.slots;
    out12[9] = rp.n_static; out12[10] = rp.max_chunks; out12[11] = rp.k_lo;
}

}  // extern "C"
