// Where to cut the global sorted order of a key array that is spread over `world` ranks, so that
// every rank ends up with its share of the whole: host-side arithmetic on digit histograms, no
// device code.  The sharded sort (mrs_dist.hip) drives it with histograms computed on the GPUs
// and gathered over RCCL; the CPU tests drive it with numpy histograms.
//
// The reference sorts on one device only (src/main.rs:594-665); this is the arithmetic of its
// scale-out, "shard by top-digit bucket", made safe for any key distribution: a cut that would
// fall too far inside a bucket sends the splitter one level down (the next digit of the keys of
// that bucket alone is histogrammed), and a cut that falls inside a run of EQUAL keys is placed
// by element order: source rank first, then position, which is the order a stable sort of the
// concatenated input defines.
#ifndef MRS_SPLIT_HPP
#define MRS_SPLIT_HPP

#include <cstdint>
#include <vector>

namespace mrs {

constexpr int kSplitTopBits = 12;  // digit width of the first histogram (the bucket table of the partition)
constexpr int kSplitSubBits = 8;   // digit width of every later one
constexpr int kSplitMaxRanks = 16;
constexpr int kSplitMaxPrefixes = kSplitMaxRanks - 1;

class Splitter {
public:
    // `sources` ranks hold the input slices; the sorted order is cut into `dests` slices (a
    // multiple of the ranks when every rank receives its share in several pieces).  Keys are
    // compared on their low `key_bits` bits (higher bits must be zero).  A cut may miss its ideal
    // position by at most `tolerance` elements before the splitter descends a level.
    Splitter(int sources, int dests, int key_bits, uint64_t tolerance);
    Splitter(int world, int key_bits, uint64_t tolerance) : Splitter(world, world, key_bits, tolerance) {}

    // The histogram wanted next: digit = (key >> shift) & ((1 << bits) - 1), counted separately for
    // each of `prefixes`, over the keys with (key >> (shift + bits)) == prefix (one empty prefix,
    // value 0, at the first level).  Returns false when every cut is placed.
    bool next(uint32_t* shift, uint32_t* bits, std::vector<uint64_t>* prefixes) const;

    // counts[(rank * num_prefixes + prefix_index) << bits | digit], for the request next() described.
    // Returns 0, or a negative value if the histograms contradict each other.
    int feed(const uint64_t* counts);

    // ---- results (valid once next() returns false) ------------------------------------------
    int world() const { return world_; }  // source ranks
    int dests() const { return dests_; }
    int num_cuts() const { return dests_ - 1; }
    uint64_t total() const { return total_; }
    uint64_t rank_size(int rank) const { return n_[rank]; }
    // Cut r (0-based, between destination r and r + 1): keys below key(r) go to destinations <= r.
    // If exact(r), the keys EQUAL to key(r) are split by element order: source rank s sends its
    // first take(r, s) of them to destinations <= r.
    uint64_t key(int r) const { return cuts_[r].key; }
    bool exact(int r) const { return cuts_[r].exact; }
    uint64_t take(int r, int rank) const { return cuts_[r].take[rank]; }
    // Position of cut r in rank's slice once that slice is stably partitioned by key class.
    uint64_t boundary(int r, int rank) const { return cuts_[r].lt[rank] + cuts_[r].take[rank]; }
    // Elements rank `src` sends to destination slice `dst`.
    uint64_t send_count(int src, int dst) const;
    uint64_t recv_total(int dst) const;

    // Class of a key under these cuts: 2 * #{r : key >= key(r)} - #{r : exact(r) and key == key(r)}.
    // Classes grow with the key; the boundaries above cut a slice that is stably partitioned by class.
    uint32_t key_class(uint64_t key) const;
    uint32_t num_classes() const { return 2u * static_cast<uint32_t>(num_cuts()) + 1u; }
    // Bucket table of the first level: table[b] = class of every key of bucket b, or 0xFF if the
    // bucket holds a cut (then the key itself must be classified).  1 << top_bits() entries.
    uint32_t top_shift() const { return top_shift_; }
    uint32_t top_bits() const { return top_bits_; }
    void bucket_table(uint8_t* table) const;
    // start[c] = number of elements of `rank` whose class is below c, for c in [0, 256).
    void class_starts(int rank, uint32_t* start) const;
    int levels_fed() const { return level_; }

private:
    struct Cut {
        uint64_t target = 0;  // ideal number of elements below the cut, over all ranks
        bool placed = false;
        bool exact = false;
        uint64_t key = 0;
        uint64_t prefix = 0;            // while descending: key >> (shift + bits) of the bucket the cut lies in
        uint64_t below = 0;             // elements of all ranks below that bucket
        std::vector<uint64_t> lt;       // per rank: elements below the bucket, finally below `key`
        std::vector<uint64_t> eq;       // per rank: elements equal to `key` (exact cuts)
        std::vector<uint64_t> take;     // per rank: how many of those go below the cut
    };
    void level_geometry(int level, uint32_t* shift, uint32_t* bits) const;
    void place_all_when_empty();

    int world_, dests_, key_bits_;
    uint64_t tolerance_;
    uint32_t top_shift_, top_bits_;
    int level_ = 0;
    bool done_ = false;
    uint64_t total_ = 0;
    std::vector<uint64_t> n_;
    std::vector<Cut> cuts_;
};

}  // namespace mrs

#endif
