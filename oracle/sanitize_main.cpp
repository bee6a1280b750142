// Host-side sanitizer run (test infrastructure only): the CPU checker (mrs_oracle.c) and the
// splitter of the sharded sort (csrc/mrs_split.hpp, pure host C++) built with
// -fsanitize=address,undefined and driven through the inputs the test suite uses.  The reference
// has latent races and unchecked indexing in the code these two restate
// (src/shaders/radix_sort_block_sum_small.wgsl:60-62 reads a word other lanes are overwriting;
// src/shaders/radix_sort_block_sum_large.wgsl:155-159 has no bounds checks), so the restatements
// are run once under the sanitizers: any report ends the program with a non-zero status
// (-fno-sanitize-recover).  Built and run by `make -C oracle sanitize`.
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
#include "mrs_oracle.h"
}
#include "mrs_split.hpp"

// what mrs_split.hip expects from the rest of the library
namespace mrs {
int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fputc('\n', stderr);
    return code;
}
void clear_error() {}
}  // namespace mrs

#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            std::fprintf(stderr, "sanitize_main: %s failed (line %d)\n", #cond, __LINE__); \
            std::exit(2);                                                             \
        }                                                                             \
    } while (0)

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64()
{
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return g_rng;
}

// ---- the checker ----------------------------------------------------------------------------
static void oracle_sorts()
{
    const uint64_t sizes[] = {0, 1, 2, 255, 256, 257, 1000, 3000, 4097};
    const int shapes[][2] = {{1, 0}, {1, 1}, {2, 1}, {2, 4}};
    for (uint64_t n : sizes) {
        for (const auto& sh : shapes) {
            const int kw = sh[0], pw = sh[1];
            for (uint32_t bits : {4u, 8u}) {
                for (uint32_t tile : {256u, 1024u}) {
                    std::vector<uint32_t> ka(n * kw + 1), kb(n * kw + 1), va(n * pw + 1), vb(n * pw + 1);
                    for (uint64_t i = 0; i < n * kw; ++i) ka[i] = static_cast<uint32_t>(next_u64()) & (i % 3 ? 0xFFFFFFFFu : 0x00FF00FFu);
                    for (uint64_t i = 0; i < n * pw; ++i) va[i] = static_cast<uint32_t>(i);
                    std::vector<uint32_t> rk(n * kw + 1), rv(n * pw + 1);
                    const uint32_t end = 32u * kw;
                    orc_stable_sort_pairs(ka.data(), pw ? va.data() : nullptr, rk.data(), pw ? rv.data() : nullptr, n, kw, pw, 0, end);
                    const uint64_t W = (n + tile - 1) / tile;
                    std::vector<uint32_t> hist((static_cast<uint64_t>(1) << bits) * (W ? W : 1) + 1);
                    const int in_b = orc_lsd_sort(ka.data(), pw ? va.data() : nullptr, kb.data(), pw ? vb.data() : nullptr, n, kw, pw,
                                                  bits, 0, end, tile, hist.data());
                    const uint32_t* ok = in_b ? kb.data() : ka.data();
                    const uint32_t* ov = in_b ? vb.data() : va.data();
                    CHECK(std::memcmp(ok, rk.data(), n * kw * 4) == 0);
                    if (pw) CHECK(std::memcmp(ov, rv.data(), n * pw * 4) == 0);
                    CHECK(orc_count_unsorted(ok, n, kw, kw == 2 ? 8 : 4) == 0);
                }
            }
        }
    }
}

static void oracle_scans_and_tests()
{
    CHECK(orc_selftest_workgroup_local_index(50) == 0);
    CHECK(orc_selftest_workgroup_calculation() == 0);
    CHECK(orc_selftest_digit6() == 0);
    CHECK(orc_selftest_digit8() == 0);
    for (uint64_t len : {256ull, 512ull, 65536ull, 65536ull + 256ull, 3ull * 65536ull}) {
        std::vector<uint32_t> a(len), b(len);
        for (uint64_t i = 0; i < len; ++i) a[i] = b[i] = static_cast<uint32_t>(next_u64() % 7);
        orc_exclusive_scan(a.data(), len);
        orc_hierarchical_scan(b.data(), len);
        CHECK(a == b);
    }
    uint64_t sizes[8];
    for (uint64_t n : {1ull, 1024ull, 1025ull, 3000ull, 1ull << 20, 1ull << 28}) CHECK(orc_prefix_buffer_sizes(n, sizes, 8) >= 0);
}

static void oracle_encoders()
{
    const uint64_t nv = 1000, nt = 3000;
    std::vector<uint32_t> blob(ORC_MORTON_BLOB_BYTES / 4);
    for (auto& w : blob) w = static_cast<uint32_t>(next_u64());
    float* f = reinterpret_cast<float*>(blob.data());
    f[51200 / 4] = 1.0f;
    for (int c = 0; c < 3; ++c) f[51216 / 4 + c] = 0.0f;
    f[51228 / 4] = 100.0f;
    for (int c = 0; c < 3; ++c) f[51232 / 4 + c] = 500.0f;
    std::vector<float> verts(nv * 12);
    for (auto& x : verts) x = static_cast<float>(next_u64() >> 40) * (1.0f / 16777216.0f);
    std::vector<uint32_t> tris(nt * 4);
    for (uint64_t i = 0; i < nt * 4; ++i) tris[i] = static_cast<uint32_t>(next_u64() % (nv + 3));  // a few out-of-range indices: clamped
    std::vector<uint32_t> codes(nt * 2);
    orc_morton_encode_lut(blob.data(), verts.data(), nv, tris.data(), nt, codes.data());
    const uint64_t n = 70001;
    std::vector<float> xyz(n * 3);
    orc_gen_points(7, n, xyz.data());
    float mn[3], mx[3];
    orc_minmax3(xyz.data(), n, mn, mx);
    std::vector<uint32_t> keys(n), idx(n);
    orc_morton_encode30(xyz.data(), n, mn, mx, keys.data(), idx.data());
    for (uint64_t i = 0; i < n; ++i) CHECK(keys[i] < (1u << 30) && idx[i] == i);
}

// ---- the splitter on simulated ranks -----------------------------------------------------------
static std::vector<uint64_t> make_slice(int dist, uint64_t n, int key_bits)
{
    const uint64_t top = key_bits >= 64 ? ~0ull : (1ull << key_bits) - 1;
    std::vector<uint64_t> k(n);
    const uint64_t hot = next_u64() & top;
    for (uint64_t i = 0; i < n; ++i) {
        switch (dist) {
            case 0: k[i] = next_u64() & top; break;
            case 1: k[i] = 12345 & top; break;
            case 2: k[i] = top; break;
            case 3: k[i] = (next_u64() % 10 < 7) ? hot : (next_u64() & top); break;
            default: k[i] = (top & ~0xFFull) + (next_u64() & 0xFF); k[i] = std::min(k[i], top); break;
        }
    }
    return k;
}

static void splitter_case(int world, int dests_per_rank, int key_bits, int dist, uint64_t tolerance)
{
    using namespace mrs;
    std::vector<std::vector<uint64_t>> slices(world);
    for (int r = 0; r < world; ++r) slices[r] = make_slice(dist, (r == 1) ? 0 : 500 + next_u64() % 3000, key_bits);
    const int dests = world * dests_per_rank;
    Splitter sp(world, dests, key_bits, tolerance);
    uint32_t shift = 0, bits = 0;
    std::vector<uint64_t> prefixes;
    int guard = 0;
    while (sp.next(&shift, &bits, &prefixes)) {
        CHECK(++guard < 16);
        const size_t np = prefixes.size();
        std::vector<uint64_t> counts((static_cast<size_t>(world) * np) << bits, 0);
        for (int r = 0; r < world; ++r)
            for (uint64_t key : slices[r])
                for (size_t j = 0; j < np; ++j) {
                    const uint32_t above = shift + bits;
                    const uint64_t p = above >= 64 ? 0 : key >> above;
                    if (sp.levels_fed() == 0 || p == prefixes[j])
                        counts[((static_cast<size_t>(r) * np + j) << bits) | ((key >> shift) & ((1ull << bits) - 1))]++;
                }
        CHECK(sp.feed(counts.data()) == 0);
    }
    uint64_t total = 0;
    for (int r = 0; r < world; ++r) total += slices[r].size();
    CHECK(sp.total() == total);
    for (int r = 0; r < world; ++r) {
        uint64_t sent = 0;
        for (int d = 0; d < dests; ++d) sent += sp.send_count(r, d);
        CHECK(sent == slices[r].size());
        uint32_t starts[256];
        sp.class_starts(r, starts);
        CHECK(starts[0] == 0);
        for (uint64_t key : slices[r]) CHECK(sp.key_class(key) < sp.num_classes());
    }
    uint64_t received = 0;
    for (int d = 0; d < dests; ++d) {
        const uint64_t got = sp.recv_total(d);
        received += got;
        CHECK(got <= total / dests + 2 * tolerance + 2);
    }
    CHECK(received == total);
    std::vector<uint8_t> table(static_cast<size_t>(1) << sp.top_bits());
    sp.bucket_table(table.data());
}

static void splitter_cases()
{
    const int worlds[] = {1, 2, 3, 8, 16};
    for (int world : worlds)
        for (int per : {1, 2})
            for (int key_bits : {13, 30, 32, 40, 64})
                for (int dist = 0; dist < 5; ++dist) {
                    if (world * per > mrs::kSplitMaxRanks) continue;
                    splitter_case(world, per, key_bits, dist, dist % 2 ? 1 : 16);
                }
}

int main()
{
    oracle_scans_and_tests();
    oracle_sorts();
    oracle_encoders();
    splitter_cases();
    std::puts("sanitize ok: checker and splitter ran clean under address + undefined-behaviour sanitizers");
    return 0;
}
