// Host-only: the Splitter (see mrs_split.hpp) and its C ABI (mrs_splitter_*, include/mrs.h).
#include "mrs_split.hpp"

#include <algorithm>
#include <cstring>

#include "mrs_internal.h"

namespace mrs {

Splitter::Splitter(int sources, int dests, int key_bits, uint64_t tolerance)
    : world_(sources), dests_(dests), key_bits_(key_bits), tolerance_(tolerance), n_(sources, 0),
      cuts_(dests > 0 ? dests - 1 : 0)
{
    const int world = sources;
    top_bits_ = key_bits < kSplitTopBits ? key_bits : kSplitTopBits;
    top_shift_ = key_bits - top_bits_;
    for (Cut& c : cuts_) {
        c.lt.assign(world, 0);
        c.eq.assign(world, 0);
        c.take.assign(world, 0);
    }
    if (cuts_.empty()) done_ = false;  // one rank: the first histogram is still wanted for the rank sizes
}

void Splitter::level_geometry(int level, uint32_t* shift, uint32_t* bits) const
{
    uint32_t s = top_shift_, b = top_bits_;
    for (int l = 0; l < level; ++l) {
        b = s < static_cast<uint32_t>(kSplitSubBits) ? s : kSplitSubBits;
        s -= b;
    }
    *shift = s;
    *bits = b;
}

bool Splitter::next(uint32_t* shift, uint32_t* bits, std::vector<uint64_t>* prefixes) const
{
    if (done_) return false;
    level_geometry(level_, shift, bits);
    prefixes->clear();
    if (level_ == 0) {
        prefixes->push_back(0);
        return true;
    }
    for (const Cut& c : cuts_)
        if (!c.placed) prefixes->push_back(c.prefix);
    std::sort(prefixes->begin(), prefixes->end());
    prefixes->erase(std::unique(prefixes->begin(), prefixes->end()), prefixes->end());
    return true;
}

int Splitter::feed(const uint64_t* counts)
{
    if (done_) return -1;
    uint32_t shift, bits;
    std::vector<uint64_t> prefixes;
    next(&shift, &bits, &prefixes);
    const size_t np = prefixes.size();
    const size_t radix = static_cast<size_t>(1) << bits;
    auto at = [&](int rank, size_t j, size_t d) { return counts[((static_cast<size_t>(rank) * np + j) << bits) + d]; };

    if (level_ == 0) {
        total_ = 0;
        for (int s = 0; s < world_; ++s) {
            uint64_t n = 0;
            for (size_t d = 0; d < radix; ++d) n += at(s, 0, d);
            n_[s] = n;
            total_ += n;
        }
        for (int r = 0; r < num_cuts(); ++r) {
            // floor((r + 1) * N / P) without overflow: N < 2^36 in practice, P <= 16
            cuts_[r].target = static_cast<uint64_t>((static_cast<unsigned __int128>(total_) * (r + 1)) / dests_);
        }
        if (total_ == 0) {
            for (Cut& c : cuts_) c.placed = true;
            done_ = true;
            ++level_;
            return 0;
        }
    }
    const bool leaf = shift == 0;
    std::vector<uint64_t> g(radix);
    for (Cut& c : cuts_) {
        if (c.placed) continue;
        const size_t j = std::lower_bound(prefixes.begin(), prefixes.end(), c.prefix) - prefixes.begin();
        for (size_t d = 0; d < radix; ++d) {
            uint64_t sum = 0;
            for (int s = 0; s < world_; ++s) sum += at(s, j, d);
            g[d] = sum;
        }
        // the digit whose bucket holds the target: below + before <= target < below + before + g[d]
        uint64_t before = 0;
        size_t dstar = radix;
        for (size_t d = 0; d < radix; ++d) {
            if (c.target < c.below + before + g[d]) { dstar = d; break; }
            before += g[d];
        }
        if (dstar == radix || c.target < c.below + before) return -2;  // the histograms do not add up
        const uint64_t lo_err = c.target - (c.below + before);
        const uint64_t hi_err = c.below + before + g[dstar] - c.target;
        const uint64_t bucket = (c.prefix << bits) + dstar;
        // a cut after the last bucket of the key space has no key value: never snap up there
        const bool top = (key_bits_ - shift) < 64 ? (bucket + 1) == (1ull << (key_bits_ - shift)) : (bucket + 1) == 0;
        const bool can_lo = lo_err <= tolerance_;
        const bool can_hi = hi_err <= tolerance_ && !top;
        auto add_lt = [&](size_t upto) {  // digits [0, upto) lie below the cut
            for (int s = 0; s < world_; ++s)
                for (size_t d = 0; d < upto; ++d) c.lt[s] += at(s, j, d);
        };
        if (can_lo && (!can_hi || lo_err <= hi_err)) {
            c.key = bucket << shift;
            add_lt(dstar);
            c.placed = true;
        } else if (can_hi) {
            c.key = (bucket + 1) << shift;
            add_lt(dstar + 1);
            c.placed = true;
        } else if (leaf) {
            c.key = bucket;
            c.exact = true;
            add_lt(dstar);
            uint64_t remaining = lo_err;  // equal keys that go below the cut, dealt in rank order
            for (int s = 0; s < world_; ++s) {
                c.eq[s] = at(s, j, dstar);
                c.take[s] = remaining < c.eq[s] ? remaining : c.eq[s];
                remaining -= c.take[s];
            }
            c.placed = true;
        } else {
            c.prefix = bucket;
            c.below += before;
            add_lt(dstar);
        }
    }
    ++level_;
    done_ = true;
    for (const Cut& c : cuts_) done_ = done_ && c.placed;
    if (done_) {
        // cuts must not cross: keys ascend, and so does every rank's boundary
        for (int r = 1; r < num_cuts(); ++r) {
            if (cuts_[r].key < cuts_[r - 1].key) return -3;
            for (int s = 0; s < world_; ++s)
                if (boundary(r, s) < boundary(r - 1, s)) return -3;
        }
        for (int r = 0; r < num_cuts(); ++r)
            for (int s = 0; s < world_; ++s)
                if (boundary(r, s) > n_[s]) return -3;
    }
    return 0;
}

uint64_t Splitter::send_count(int src, int dst) const
{
    const uint64_t hi = dst == dests_ - 1 ? n_[src] : boundary(dst, src);
    const uint64_t lo = dst == 0 ? 0 : boundary(dst - 1, src);
    return hi - lo;
}

uint64_t Splitter::recv_total(int dst) const
{
    uint64_t t = 0;
    for (int s = 0; s < world_; ++s) t += send_count(s, dst);
    return t;
}

uint32_t Splitter::key_class(uint64_t key) const
{
    uint32_t c = 0;
    for (const Cut& cut : cuts_) {
        if (key >= cut.key) c += 2;
        if (cut.exact && key == cut.key) c -= 1;
    }
    return c;
}

void Splitter::bucket_table(uint8_t* table) const
{
    const uint64_t buckets = 1ull << top_bits_;
    for (uint64_t b = 0; b < buckets; ++b) {
        const uint64_t lo = b << top_shift_;
        const uint64_t hi = lo + ((1ull << top_shift_) - 1);
        bool holds_cut = false;
        for (const Cut& c : cuts_) {
            if (c.exact ? (c.key >= lo && c.key <= hi) : (c.key > lo && c.key <= hi)) holds_cut = true;
        }
        table[b] = holds_cut ? static_cast<uint8_t>(kMapBySplitters) : static_cast<uint8_t>(key_class(lo));
    }
}

void Splitter::class_starts(int rank, uint32_t* start) const
{
    uint64_t count[256] = {0};
    uint64_t pos = 0;     // first element not yet attributed to a class
    uint32_t cuts_seen = 0;
    int r = 0;
    while (r < num_cuts()) {
        // the group of cuts that share one key value
        int end = r;
        uint32_t exact_cuts = 0;
        uint64_t eq = 0;
        while (end < num_cuts() && cuts_[end].key == cuts_[r].key) {
            if (cuts_[end].exact) {
                ++exact_cuts;
                eq = cuts_[end].eq[rank];
            }
            ++end;
        }
        const uint64_t lt = cuts_[r].lt[rank];
        count[2 * cuts_seen] += lt - pos;  // keys between the previous cut value and this one
        cuts_seen += static_cast<uint32_t>(end - r);
        pos = lt;
        if (exact_cuts) {
            count[2 * cuts_seen - exact_cuts] += eq;  // keys equal to this cut value
            pos += eq;
        }
        r = end;
    }
    count[2 * cuts_seen] += n_[rank] - pos;
    uint64_t run = 0;
    for (int c = 0; c < 256; ++c) {
        start[c] = static_cast<uint32_t>(run);
        run += count[c];
    }
}

}  // namespace mrs

// ---- C ABI ------------------------------------------------------------------------------
struct mrs_splitter {
    mrs::Splitter impl;
    mrs_splitter(int world, int key_bits, uint64_t tol) : impl(world, key_bits, tol) {}
};

using namespace mrs;

extern "C" int mrs_splitter_create(int32_t world, int32_t key_bits, uint64_t tolerance, mrs_splitter** out)
{
    clear_error();
    if (!out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_splitter_create: null output pointer");
    *out = nullptr;
    if (world < 1 || world > kSplitMaxRanks)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_splitter_create: world must be in [1, %d]", kSplitMaxRanks);
    if (key_bits < 1 || key_bits > 64) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_splitter_create: key_bits must be in [1, 64]");
    *out = new mrs_splitter(world, key_bits, tolerance);
    return MRS_OK;
}

extern "C" int mrs_splitter_destroy(mrs_splitter* sp)
{
    clear_error();
    delete sp;
    return MRS_OK;
}

extern "C" int mrs_splitter_next(mrs_splitter* sp, uint32_t* shift, uint32_t* bits, uint32_t* num_prefixes,
                                 uint64_t* prefixes)
{
    clear_error();
    if (!sp || !shift || !bits || !num_prefixes || !prefixes)
        return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_splitter_next: null argument");
    std::vector<uint64_t> p;
    if (!sp->impl.next(shift, bits, &p)) {
        *bits = 0;
        *num_prefixes = 0;
        return MRS_OK;
    }
    *num_prefixes = static_cast<uint32_t>(p.size());
    for (size_t i = 0; i < p.size(); ++i) prefixes[i] = p[i];
    return MRS_OK;
}

extern "C" int mrs_splitter_feed(mrs_splitter* sp, const uint64_t* counts)
{
    clear_error();
    if (!sp || !counts) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_splitter_feed: null argument");
    const int rc = sp->impl.feed(counts);
    if (rc) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_splitter_feed: inconsistent histograms (code %d)", rc);
    return MRS_OK;
}

extern "C" int mrs_splitter_get(const mrs_splitter* sp, mrs_split_result* out)
{
    clear_error();
    if (!sp || !out) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_splitter_get: null argument");
    const Splitter& s = sp->impl;
    uint32_t shift, bits;
    std::vector<uint64_t> p;
    if (s.next(&shift, &bits, &p)) return fail(MRS_ERR_INVALID_ARGUMENT, "mrs_splitter_get: the splitter still wants histograms");
    std::memset(out, 0, sizeof *out);
    out->world = s.world();
    out->levels = s.levels_fed();
    out->total = s.total();
    for (int r = 0; r < s.num_cuts(); ++r) {
        out->cut_key[r] = s.key(r);
        if (s.exact(r)) out->exact_mask |= 1u << r;
    }
    for (int src = 0; src < s.world(); ++src) {
        out->rank_size[src] = s.rank_size(src);
        for (int dst = 0; dst < s.world(); ++dst) out->send_count[src][dst] = s.send_count(src, dst);
    }
    return MRS_OK;
}
