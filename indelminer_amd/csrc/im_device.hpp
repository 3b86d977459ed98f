// im_device.hpp -- shared declarations of the HIP side (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "indelminer_amd.h"

namespace im {

// Reference contigs resident in HBM, two forms:
//   ascii : the bytes as the reference keeps them (src/shared.c:46-82); each
//           contig starts 256-byte aligned with >= 64 zero bytes before and
//           after, so unaligned 4/16-byte reads at a window edge stay in bounds.
//           Needed by the diagonal scan, which compares raw bytes
//           (W[i][j] = MATCH iff i == j, src/localalign.c:61-67).
//   pk    : 2-bit codes (base2bits, src/alignment.c:11-24: A,a=0 C,c=1 G,g=2
//           T,t=3, anything else 0), 4 bases per byte, first base in the least
//           significant bits, so an unaligned dword at byte p/4 shifted right by
//           2*(p%4) starts with the k-mer at p.  >= 1 KiB of zero bytes follow the
//           last contig (the vote reads up to a sweep past a window).  Needed by
//           the k-mer band vote only.
struct RefDev {
    const uint8_t*  ascii;
    const uint8_t*  pk;
    const int64_t*  asc_off;    // [n_contigs] byte offset of contig start in ascii
    const int64_t*  pk_off;     // [n_contigs] byte offset of contig start in pk
    const int32_t*  len;        // [n_contigs]
    int32_t         n_contigs;
};

constexpr int kShortRead = 255;      // longest read of the four-positions-per-lane kernels (im_realign.hip)
constexpr int kMaxWaveGaps = 60;     // widest numgaps of the lane-per-diagonal band kernel (a band of 61 diagonals and its two borders in one wave)

struct RealignArgs {
    RefDev      ref;
    im_dev_batch batch;
    im_params   P;
    int32_t     keep_slots;     // 1: evidence slots of reads without realigned evidence are left as they are
    const int32_t* n_dev;       // device-resident batch size (null: batch.n), batch.n is then the upper bound
    int32_t     first;          // realign_kernel: the first read of this launch's slice (launch_realign cuts a batch beyond its largest grid)
};

// The read-group -> range[1] table of the insert-length hashtable, flattened into one blob:
// int32 bin_start[20] (17 used; 16 bins of the qhash of size 2^4, src/indelminer.c:702), name_off[m], name_len[m],
// range_max[m] (m = max(n,1); the entries of a bin in chain order, head first), then the names.
struct RgTable {
    const uint8_t* blob;
    int32_t n;
    int32_t bytes;
};

size_t triage_scratch_bytes(int32_t n_records);
size_t triage_scratch_zero_offset(int32_t n_records, size_t* bytes);
hipError_t launch_triage(const RefDev& ref, const RgTable& rg, int32_t* depth_diff, const im_triage_params& tp,
                         const im_dev_records& recs, const im_dev_cands& out, void* scratch, hipStream_t stream);

hipError_t launch_flush_cut(const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                            int32_t a0, int32_t a1, int32_t b0, int32_t b1_end, int32_t marker, int32_t flush_id,
                            uint64_t* cut_word, const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap, hipStream_t stream);
hipError_t launch_flush_seq(const im_flush_desc* desc, int32_t n_fl, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                            int32_t* consumed, const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap, int32_t pe_base,
                            int32_t pe_count, hipStream_t stream);
size_t groupby_scratch_bytes(int32_t n_slots);
hipError_t launch_groupby_init(int32_t n_slots, void* scratch, hipStream_t stream);
hipError_t launch_groupby(int32_t n_layout, int32_t n_slots, const int32_t* n_cand_dev, const int32_t* cls, const int32_t* b1, const int32_t* b2, const int32_t* consumed,
                          int32_t tie_desc, int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count,
                          int32_t* counts, void* scratch, hipStream_t stream);
// im_flushwide.hip: the flush list and the group-by chip-wide (three launches)
size_t flushgroup_scratch_bytes(int32_t n_slots, int32_t n_fl_cap);
hipError_t launch_flushgroup_init(int32_t n_slots, int32_t n_fl_cap, void* scratch, hipStream_t stream);
hipError_t launch_flush_groupby(int32_t n_slots_layout, int32_t n_fl_layout, const im_flush_desc* desc, int32_t n_fl,
                                const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                                const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap, int32_t pe_base, int32_t pe_count,
                                int32_t tie_desc, int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count, int32_t* counts,
                                void* scratch, hipStream_t stream);
hipError_t launch_depth_scan(int32_t* depth, int64_t n, int32_t* sums, hipStream_t stream);
hipError_t launch_depth_scan_tiled(int32_t* depth, int64_t n, int32_t* sums, hipStream_t stream);
hipError_t launch_depth_query_tiled(int32_t nq, const int32_t* beg, const int32_t* end, const int32_t* depth, const int32_t* sums,
                                    int64_t clen, uint32_t* out, uint32_t* out_max, hipStream_t stream);

// launchers (im_realign.hip / im_cluster.hip)
hipError_t launch_pack_reference(const uint8_t* ascii, uint64_t* pk, int64_t n_bases_padded,
                                 hipStream_t stream);
hipError_t launch_realign(const RealignArgs& a, int n_cu, hipStream_t stream);
// im_realign_long.hip: the reads of kShortRead + 1 .. IM_MAX_READ bases of the same batch (numgaps == 0)
hipError_t launch_realign_long(const RealignArgs& a, int n_cu, hipStream_t stream);

// im_realign_any.hip: the reads the laid-out kernels leave IM_ST_UNSUPPORTED (any length, any band width), one lane per read.
// pick lists them (all != 0: every read of the batch, which no other kernel has seen) into list[] with counters[0] = how many,
// [1] = the longest read, [2] = the widest window among them; the caller sizes the arena from those (n_waves waves of 64 lanes).
hipError_t launch_realign_any_pick(const RealignArgs& a, int all, int32_t* list, int32_t* counters, hipStream_t stream);
// lane_shift: log2 of the reads a wave holds at a time (6 = every lane; fewer reads per wave = more waves for a small batch)
size_t realign_any_arena_bytes(int32_t max_read, int32_t max_window, uint32_t numgaps, int32_t lane_shift, int32_t n_waves);
hipError_t launch_realign_any(const RealignArgs& a, const int32_t* list, int32_t* counters, int32_t* arena,
                              int32_t max_read, int32_t max_window, int32_t lane_shift, int32_t n_waves, hipStream_t stream);

// im_results.hip
hipError_t launch_compact_results(const im_read_result* res, int32_t n_cap, const int32_t* n_dev, int32_t* status, int32_t* slot,
                                  im_read_result* compact, int32_t* count, int n_cu, hipStream_t stream);

size_t cluster_scratch_bytes(int32_t n);
hipError_t launch_cluster_sr(int32_t n_cap, const int32_t* n_dev,
                             const int32_t* cls, const int32_t* b1, const int32_t* b2,
                             int32_t marker, int32_t tie_desc,
                             int32_t* order, int32_t* cl_first, int32_t* cl_count,
                             uint8_t* used, int32_t* n_clusters,
                             void* scratch, size_t scratch_bytes, hipStream_t stream);

hipError_t launch_cluster_records(int32_t tid, const int32_t* counts, const int32_t* order, const int32_t* cl_first,
                                  const int32_t* cl_count, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                                  int32_t* recs, int32_t cap, hipStream_t stream);
int cluster_small_max();
size_t cluster_hist_scratch_bytes(int32_t n_slots);
hipError_t launch_cluster_hist_init(int32_t n_slots, void* scratch, size_t scratch_bytes, hipStream_t stream);
hipError_t launch_cluster_hist(int32_t n_slots, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                               int32_t marker, int32_t tie_desc,
                               int32_t* order, int32_t* cl_first, int32_t* cl_count,
                               uint8_t* used, int32_t* out_counts,
                               void* scratch, size_t scratch_bytes, hipStream_t stream);
hipError_t launch_cluster_small(int32_t n_slots, const int32_t* n_slots_dev,
                                const int32_t* cls, const int32_t* b1, const int32_t* b2,
                                int32_t marker, int32_t tie_desc,
                                int32_t* order, int32_t* cl_first, int32_t* cl_count,
                                uint8_t* used, int32_t* out_counts, hipStream_t stream);

size_t gather_scratch_bytes(int32_t n);
hipError_t launch_gather_evidence(const im_read_result* res, int32_t n,
                                  int32_t* cls, int32_t* b1, int32_t* b2, int32_t* src,
                                  int32_t cap, int32_t* n_out, void* scratch, size_t scratch_bytes,
                                  hipStream_t stream);

hipError_t launch_depth_build(int64_t clen, int32_t n_seg, const int32_t* seg_start, const int32_t* seg_len,
                              int32_t* depth, int32_t* sums, hipStream_t stream);
int64_t depth_tiles(int64_t clen);
int64_t depth_sums_ints(int64_t clen);
hipError_t launch_depth_query(int32_t nq, const int32_t* beg, const int32_t* end, const int32_t* depth, int64_t clen,
                              uint32_t* out, hipStream_t stream);

// tasks within IM_MAX_SW_TARGET / IM_MAX_READ run in the LDS form; when the batch holds longer ones (big_grid > 0) a second launch
// with its boundary rows in big_scratch (support_big_scratch_bytes) takes those
size_t support_big_scratch_bytes(int64_t max_target, int64_t max_query, int32_t n_tasks, int32_t* grid_out);
hipError_t launch_support(int32_t n_tasks, const uint8_t* targets, const int64_t* t_off,
                          const uint8_t* queries, const int64_t* q_off, int32_t* out, int64_t max_target, int64_t max_query,
                          void* big_scratch, int32_t big_grid, int n_cu, hipStream_t stream);

}  // namespace im
