// Private to the device runtime (srn_runtime.hip, srn_hostpipe.hip, srn_group.hip): the per-index device state, per-call workspaces and the
// test / experiment knobs.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "srn_kernels.h"

namespace srn {

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(SRN_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Test / experiment knobs (environment variables SRN_*): read ONCE, when the library is first used, never on the launch path;
// srn_debug_reload_knobs() re-reads them (the tests switch kernel paths between calls).  All defaults = production behaviour.
struct Knobs {
    int fast_how_many_max = 64;   // SRN_FAST_HOW_MANY_MAX: the largest how_many the fast kernels take (round 6: 64 -- the threshold sample takes the ceil(n / 8)-th largest per wave; until round 5: 24)
    bool no_viol = false;   // SRN_NO_VIOL: an index with incomplete lists takes the general kernel for every query, as until round 5 (A/B)
    bool no_masks = false, no_merge = false, dense = false, no_fast = false, no_mid = false, no_big = false, no_long = false, debug = false;   // no_long (SRN_NO_LONG): without the LONG instantiation (sessions of 11..20 items go to the general kernel, as until round 4)   // no_mid (SRN_NO_MID): the launch sequence without the fast kernel's MID instantiation (what it would take goes to the general kernel, as before round 4)
    int hot_slots = -1, sketch_slots = -1, lds_budget_kb = 0, grid_mult = 16, fast_runs = 0;
    bool grid_mult_set = false;
    int host_first_pct = 0;   // SRN_HOST_FIRST_PCT (experiments): a host-pointer batch of 8 192 .. 65 536 queries in TWO chunks, the first this percentage of it (0: the default policy of srn_hostpipe.hip)
    int host_chunks = 0;      // SRN_HOST_CHUNKS: number of chunks a host-pointer batch is cut into (0 = by size, srn_hostpipe.hip)
    int copy_slices = 0;      // SRN_COPY_SLICES: slices a result block is cut into for the copy threads (0 = one per thread)
    bool host_nocopy = false; // SRN_HOST_NOCOPY (experiments): the chunked host path leaves the results in its pinned staging
    bool host_trace = false;  // SRN_HOST_TRACE (experiments): per-call timeline of the chunked host path on stderr
    bool timing = false;      // SRN_TIMING: kernel timing on from the start (srn_kernel_timing switches it per index)
    int d2h_blocks = 0;       // SRN_D2H_BLOCKS: workgroups of the chunked host path's own download kernel (0 = hipMemcpyAsync, the default: measured faster, profiles/r03_host_pipe_probe.txt)
    int tiny_max = 256;       // SRN_TINY_MAX: host-pointer batches of up to this many sessions take the zero-copy latency path
    bool no_tiny_spin = false;    // SRN_TINY_SPIN=0: the fused launch's caller waits for the stream instead of spinning on the kernel's last pinned word
    int tiny_fused_max = 48;      // SRN_TINY_FUSED_MAX: sessions per call up to which the latency path is the fused launch (one workgroup per session).  Measured (config 3, host batches): 48 per call p50 56 / p90 77 us against 78 / 83 for prep + general kernel; 64 per call 57 / 103 against 79 / 85 -- a call's chance of a second phase (a query for the general kernel: 0.3 % each) grows with its size, and the p90 with it
    bool no_tiny_fused = false;   // SRN_TINY_FUSED=0: a single-session call through five launches (prep, fast kernel, general kernel, finish, finish-big) instead of the fast kernel's one-launch TINY form
    int tiny_phases = 0;      // SRN_TINY_PHASES (experiments): the latency path's launches behind the fast kernel -- 0: all of them with every call; 1: finish + finish-big, then MID / general kernel / a second finish only for a call that listed queries for them (a second wait then); 2: finish alone in the first phase.  Measured (profiles/r05_latency_phases.txt): one query per call p50 46 us against 49 with 2, p99 71 against 63; calls of 4 and 16 queries lose at p90 (a second wait more often): 0 stays
    int tiny_fast = 2;        // SRN_TINY_FAST: the latency path's kernels -- 0: prep + general kernel (rounds 1-3); 1: the fast kernel's launch sequence only where the batch has a session of > 8 items (which puts
                              // the whole batch on the general kernel's non-position-set build: 144 us against 52 us per call on config 3); 2 (default): also for calls of <= 32 sessions (one session per call,
                              // config 3, max_items 4: p50 53.3 -> 47.8 us, p90 63.7 -> 54.1; larger rounds stay on the general kernel -- one workgroup per query runs them all at once: nothing to gain,
                              // four more launches to pay: profiles/r04_serving_tiny_fast.txt); 3: wherever the batch's shape allows it (experiments)
    int row_slots16 = -1;     // SRN_ROW_SLOTS = 16 | 64: the device row layout (-1 = by index kind: 64-byte slots unsharded, 16-byte fragment slots for item shards)
    int order_min = 131072;    // SRN_ORDER_MIN: batches of at least this many queries are served in the order of their most popular item, an eighth of the order per XCD (0 = never); the ordering
                              // pass is one radix sort of the batch's keys behind the prep kernel
    bool no_sback_finish = true;    // SRN_SBACK_FINISH=1 (experiment): the wave-per-query back end finishes rows of <= 63 entries itself instead of leaving a record for vmis_finish_kernel.  Measured: the finish kernels' share falls 0.167 -> 0.122 ms per 131 072 queries, the kernel grows 1.606 -> 1.685 (two more dependent gathers per query on a kernel bound by its requests): off
    bool no_sback_pbytes = true;    // SRN_SBACK_PBYTES=1 (experiment): presence bytes in the neighbours pipeline's exchange records -- the fronting rank marks, per neighbour, which shards hold a fragment of it (the shards' bitmaps all-gathered at set_postings); a back end asks only for fragments that exist: half the requests at G = 8, no look-up of its own.  Measured in round 5: kernel 1.61 -> 1.95 ms (its 22 byte loads were waited for one by one); round 6, read in one batch and the absent lanes truly silent: 1.149 against 1.149 ms, front 0.30 -> 0.36 -- half the fragment requests buy nothing, the kernel's time does not follow them (profiles/r06_sback_ab.txt)
    bool no_sback_second = false;   // SRN_NO_SBACK_SECOND (experiments): what the wave-per-query back end cannot hold goes straight to the general kernel (no fast-kernel back end over the list)
    bool no_sback = false;    // SRN_NO_SBACK: the shard group's back end through vmis_fast_kernel's FM_BACK instantiation (rounds 4) instead of the wave-per-query kernel of srn_sback.hip
    bool sback_bitmap = false;     // SRN_SBACK_BITMAP=1 (experiments): that kernel asks its presence bitmap before it fetches a fragment.  Measured on config 3 cut in 8: half the fragment
                                   // fetches, but one more DEPENDENT round trip per query on a kernel that spends 65 % of its time waiting for memory -- 1.65 ms with, 1.52 ms without
    double xgmi_gbps = 76.8;       // SRN_XGMI_GBPS: what one xGMI link moves per direction (AUTO's input below)
    int sback_stream_mode = -1;    // SRN_SBACK_STREAM: 1 = the streaming form wherever the shards have it, 0 = never, unset = AUTO (round 6): a group with real peers (RCCL / callbacks) takes it
                                   // unless its exchanges overlap the previous batch (srn_shard_group_set_overlap) -- it ships a third of the gather form's bytes and costs 0.4 ms more compute per
                                   // rank and batch, which pays as soon as the exchange is on the batch's critical path (bench.py: item_sharded.local_g8.exchange_model); an in-process group's
                                   // "exchange" is a device copy: gather
    bool no_sback_stream = false;  // (= sback_stream_mode == 0)  SRN_SBACK_STREAM=1 (experiments) turns the STREAMING form of that kernel on: the shard keeps its fragments a second time in posting order (8 B per posting), the
                                   // exchange carries the neighbours as positions in the posting lists (a third of the bytes), the back end reads the lists' kept prefixes coalesced.  Built and
                                   // measured in round 5 (profiles/r05_sback_stream_ab.txt): a query's ~4 750 kept postings are 3.5 x its ~1 360 neighbours, and the walk, which memory no longer
                                   // bounds, is issue-bound on them: 1.92 ms per 131 072 queries against 1.54 for the gather form.  Off by default; same rows either way.
    int sback_min_shards = 8; // SRN_SBACK_MIN_SHARDS: shards of an index cut in at least this many get the frag8 rows (below: fragments of > 4 items are common and the 1 024 + 1 024-word
                              // geometry too small -- config 3 cut in 4 handed 117 K of 131 K queries on; the FM_BACK form of the fast kernel serves those groups)
    int lanes = 4;            // SRN_PREDICT_LANES: concurrent rounds of the srn_predict combiner (srn_combine.cpp)
    bool geometry_default() const { return !no_masks && !no_merge && !dense && hot_slots < 0 && sketch_slots < 0 && lds_budget_kb == 0; }
};
Knobs knobs();   // (a copy: the tests re-read the environment between calls)

struct HostPipe;   // srn_hostpipe.hip: staging rings + streams of the chunked host-pointer path

struct Workspace {
    hipStream_t stream = nullptr;   // own stream for host-pointer calls
    static constexpr int RING = 64;               // per-call events: start, after main kernel, after retry pass, after prep kernel
    hipEvent_t ev[RING][5] = {};                  // ... [4] = after the fast kernel (== [3] when the launch did not use it)
    bool ring_timed[RING] = {};                   // the call recorded all five (kernel timing on: srn_kernel_timing); otherwise only [2], the end of the call
    uint64_t calls = 0, untimed_calls = 0; uint32_t last_retry = 0, last_nq = 0;
    bool last_fast = false;      // the last call went through the fast kernel: h_retry[1] = what it handed to the general kernel (otherwise: all of last_nq)
    bool last_mid = false;       // ... and through its MID instantiation: h_retry[2] = the queries the lean instantiation listed for it
    bool last_untimed = false;   // the last call took the latency path: no events were recorded for it
    // device scratch
    uint32_t* retry_list = nullptr; size_t retry_cap = 0; uint32_t* retry_cnt = nullptr;
    char* gscratch = nullptr; size_t gscratch_bytes = 0;
    char* spill = nullptr; size_t spill_bytes = 0;   // per-block global copies of the neighbour lists
    char* prep = nullptr; size_t prep_bytes = 0;     // per-query records of the prep kernel
    char* order = nullptr; size_t order_bytes = 0;   // the batch's order keys as the prep kernel wrote them | sorted | the sort's scratch
    char* sb_scr = nullptr; size_t sb_scr_bytes = 0; // the streaming back end's per-wave scratch
    char* order2 = nullptr; size_t order2_bytes = 0; // ... of the second record set (SRN_FLAG_INPUTS_RESIDENT: call i + 1's prep kernel and sort run beside call i's kernels, which still read theirs)
    uint32_t* retry_list2 = nullptr; size_t retry_cap2 = 0; uint32_t* retry_cnt2 = nullptr;   // what the second LDS tier could not hold either
    uint32_t* slow_list = nullptr; size_t slow_cap = 0; uint32_t* slow_cnt = nullptr;          // what the fast kernel hands to the general one
    char* fin = nullptr; size_t fin_bytes = 0;   // records for vmis_finish_kernel
    char* big = nullptr; size_t big_bytes = 0;   // overflow entries + list for vmis_finish_big_kernel
    char* pin = nullptr; size_t pin_bytes = 0;   // pinned, device-mapped staging of the latency path (a handful of queries on host pointers)
    // staging for host-pointer calls
    char* stage = nullptr; size_t stage_bytes = 0;
    std::mutex call_mu;            // a stream-bound workspace serves one call at a time: two host threads enqueueing on the SAME stream must not interleave their launch sequences (they share these buffers)
    bool cnt_dirty = true;          // slow_cnt may hold non-zero words (any launch sequence but a fused one that handed nothing on leaves them so)
    uint32_t tiny_seq = 0;          // number of the latency path's last fused call (the kernel writes it into h_retry[5] when the row is complete)
    uint32_t* h_retry = nullptr;   // pinned
    uint32_t* h_retry_dev = nullptr;   // ... as the device sees it (vmis_finish_big_kernel writes the two counters there)
    bool h_retry_valid = false; hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // the global-table pass forked beside the finish kernels (srn_runtime.hip)
    hipEvent_t ev_block = nullptr; // blocking-sync event of the latency path (rounds shared by several callers)
    // SRN_FLAG_INPUTS_RESIDENT: this call's prep kernel on a side stream, beside the previous call's kernels (two sets of prep records)
    hipStream_t side = nullptr; hipEvent_t ev_prep[2] = {}, ev_done[2] = {}; char* prep2 = nullptr; size_t prep2_bytes = 0; uint64_t resident_calls = 0; bool rec_used[2] = {false, false};
};

struct ServeState;   // the persistent latency path's resident workgroups (srn_runtime.hip, "serve")
struct DeviceState {
    int device = 0;
    std::atomic<ServeState*> serve{nullptr}; std::vector<ServeState*> serve_retired;   // (retired: stopped, kept until device_release -- a concurrent srn_predict may still hold the pointer)
    std::vector<void*> allocs; uint64_t bytes = 0;
    DeviceIndex di{};
    ItemMeta* d_meta = nullptr;
    FastParams fast{};            // packed row slots + idf bounds of the fast kernel (row_packed == nullptr: no fast path for this index)
    std::atomic<uint64_t> sback_launches{0};
    void* sb_frag_post = nullptr; const uint32_t* sb_post_for = nullptr; uint64_t sb_frag_post_bytes = 0;   // the fragments in the posting order of the replicated lists at sb_post_for (device_sback_attach_postings)
    size_t sback_present_words = 0;
    SBackParams sback{};          // item shards: frag8 rows + presence bitmap of the wave-per-query back end (frag8 == nullptr: the FM_BACK form of the fast kernel serves)
    uint32_t host_max_row_len = 0;
    int n_cu = 256;
    int lds_per_block_max = 65536;
    std::mutex mu; std::vector<Workspace*> free_ws; std::vector<Workspace*> all_ws;
    std::vector<std::pair<void*, Workspace*>> stream_ws;   // device-pointer calls: one workspace per user stream
    Workspace* last_ws = nullptr;   // for srn_last_kernel_ms (single-threaded measurement use)
    std::vector<HostPipe*> free_pipes, all_pipes;   // chunked host-pointer batches (srn_hostpipe.hip), pooled like the workspaces
    unsigned long long* d_phase = nullptr; bool phase_on = false;   // debug per-phase cycle counters
    std::atomic<bool> timing{false};   // record the per-kernel events of every call (srn_kernel_timing; SRN_TIMING=1): each event costs ~6 us of idle stream
};

// ---- shared helpers (srn_runtime.hip) ----
int ensure(char** p, size_t* have, size_t need);   // grow-only device scratch
Workspace* ws_acquire(DeviceState* d, bool bind_to_stream, void* user_stream);
void ws_release(DeviceState* d, Workspace* w, bool bound);
void hostpipes_free(DeviceState* d);   // srn_hostpipe.hip
int device_predict_host_pipelined(DeviceState* d, const FlatIndex& ix, const LaunchParams& p, const uint64_t* h_items, const uint32_t* h_qoff,
                                  uint64_t* h_ids, double* h_scores, uint32_t* h_counts);
uint32_t hostpipe_chunks(uint32_t nq, uint32_t how_many);

}  // namespace srn
