// libripp_hip.so -- engine + C ABI (include/ripp_hip.h).  One process drives ONE MI355X (one process per GPU);
// multi-GPU proofs are sharded by index residue and combined by the caller over RCCL (ripp_amd/sharded.py).
//
// There is no CPU compute fallback in this file: every data-parallel step is a HIP kernel from kernels.hpp.  The
// host does only what the reference's host code does between its parallel sections -- Fiat-Shamir hashing, the
// per-round final exponentiation of ONE Fp12 value, scalar inversion -- exactly the serial glue of
// sipp/src/lib.rs:56-60,80-85,94.
#include <hip/hip_runtime.h>
#include <atomic>
#include <csignal>
#include <sched.h>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <future>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#ifndef RIPP_BUILD_STAMP
#define RIPP_BUILD_STAMP __DATE__ " " __TIME__       // (the Makefile passes a hash of the sources)
#endif
#include "../../include/ripp_hip.h"
#include "kernels.hpp"
#include "line_products.hpp"
#include "msm.hpp"
#include "scale.hpp"
#include "vm.hpp"
#include "fq_curve.hpp"
#include "fq_curve2.hpp"
#include "fq_line_products.hpp"
#include "fq_line_products_k.hpp"
#include "fq_miller.hpp"
#include "fq_scale.hpp"
#include "fq_msm.hpp"
#include "vm_fold2.hpp"
#include "host_fs.hpp"
#include "wire.hpp"

using namespace ripp;

static_assert(sizeof(ripp_fp) == sizeof(Fp) && sizeof(ripp_fr) == sizeof(Fr), "limb layout");
static_assert(sizeof(ripp_gt) == sizeof(Fp12) && sizeof(ripp_g1a) == sizeof(G1A) && sizeof(ripp_g2a) == sizeof(G2A), "layout");
static_assert(sizeof(ripp_g1j) == sizeof(G1J) && sizeof(ripp_g2j) == sizeof(G2J), "layout");

namespace {

std::mutex g_mu;
ripp_config g_cfg{}; bool g_cfg_set = false;      // ripp_configure(): process-wide, applied by Engine::refresh_switches at the start of every C-ABI call
thread_local std::string g_err;          // last error message of the CALLING thread (errno-style; ripp_last_error)

// Persistent host workers for the per-round serial glue (final exponentiations, GT powers, KZG quotients).  std::async spawns a
// thread per call, which costs ~0.1 ms and occasionally 1-2 ms -- on the critical path of every one of the 20 rounds of a proof.
#if defined(__HIP_DEVICE_COMPILE__)
#define RIPP_CPU_RELAX() ((void)0)
#else
#define RIPP_CPU_RELAX() __builtin_ia32_pause()
#endif
class HostPool {
public:
    explicit HostPool(unsigned n) { for (unsigned i = 0; i < n; ++i) th_.emplace_back([this]() { run(); }); }
    ~HostPool() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); for (auto& t : th_) t.join(); }
    unsigned size() const { return (unsigned)th_.size(); }
    template <class F> auto submit(F&& f) -> std::future<decltype(f())> {
        auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::forward<F>(f));
        auto fut = task->get_future();
        push([task]() { (*task)(); });
        return fut;
    }
    // While `hot`, idle workers poll instead of sleeping (a futex wake-up costs 0.1-0.2 ms, as much as the tasks of a round's host phase).
    // Off by default and during the statement hash: a polling sibling hyper-thread slows the hashing core down.
    void set_hot(bool h) { hot_.store(h, std::memory_order_release); }
    // fn(0) .. fn(n - 1), fn(0) on the caller's thread; returns when all are done.  The waits SPIN (the tasks are 0.1-0.6 ms pieces of a
    // round's serial host phase, shorter than a futex sleep + wake-up): workers poll for ~2 ms after their last task before they block.
    template <class F> void parallel(int n, F&& fn) {
        if (n <= 0) return;
        std::atomic<int> left{n - 1};
        for (int t = 1; t < n; ++t) push([&fn, &left, t]() { fn(t); left.fetch_sub(1, std::memory_order_release); });
        fn(0);
        while (left.load(std::memory_order_acquire) > 0) RIPP_CPU_RELAX();
    }
private:
    void push(std::function<void()> job) {
        { std::lock_guard<std::mutex> lk(mu_); q_.emplace_back(std::move(job)); }
        pending_.fetch_add(1, std::memory_order_release);
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_one();
    }
    void run() {
        for (;;) {
            std::function<void()> job;
            const auto t0 = std::chrono::steady_clock::now();
            for (int spin = 0; hot_.load(std::memory_order_acquire) && pending_.load(std::memory_order_acquire) == 0; ++spin) {      // hot wait
                RIPP_CPU_RELAX();
                if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
            }
            {
                std::unique_lock<std::mutex> lk(mu_);
                if (q_.empty()) { sleepers_.fetch_add(1); cv_.wait(lk, [this]() { return stop_ || !q_.empty(); }); sleepers_.fetch_sub(1); }
                if (stop_ && q_.empty()) return;
                job = std::move(q_.front()); q_.pop_front();
                pending_.fetch_sub(1, std::memory_order_release);
            }
            job();
        }
    }
    std::vector<std::thread> th_; std::deque<std::function<void()>> q_; std::mutex mu_; std::condition_variable cv_; bool stop_ = false;
    std::atomic<int> pending_{0}, sleepers_{0}; std::atomic<bool> hot_{false};
};
static int effective_cpus();
// 6 workers (+ the caller's thread); 11 where the process may use 14 CPUs or more: the pipelined tail then cuts each of its six final exponentiations in two
// bit ranges (job_tail_values) -- 12 tasks of ~0.65 ms instead of 6 of ~0.95 ms on the critical path of the last 12 rounds
HostPool& host_pool() {      // (RIPP_HOST_WORKERS: A/B override, read once when the pool is created)
    static HostPool pool([]() -> unsigned { if (const char* s = std::getenv("RIPP_HOST_WORKERS")) { const int v = std::atoi(s); if (v >= 1 && v <= 64) return (unsigned)v; } return effective_cpus() >= 14 ? 11u : 6u; }());
    return pool;
}
// CPUs this PROCESS TREE may use: the affinity mask capped by the cgroup's CPU quota (cpu.max: the GPU pool's boxes give 16).  The ranks of a sharded proof
// are separate processes on ONE node and share that allowance.
static int effective_cpus() {
    static const int n = []() {
        int c = (int)std::thread::hardware_concurrency();
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) c = a; }
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0}; long period = 0;
            if (std::fscanf(f, "%31s %ld", q, &period) == 2 && period > 0 && q[0] != 'm') { const long quota = std::atol(q); if (quota > 0) c = std::min<long>(c, (quota + period - 1) / period); }
            std::fclose(f);
        }
        return std::max(1, c);
    }();
    return n;
}
// Polling ("hot") workers are worth it only when every rank on the node has cores to spare for them: 6 workers + the prover's thread per rank.  With fewer
// (8 ranks under a 16-CPU quota: two CPUs per rank) spinning threads would time-slice against the other ranks' critical host phases.
static bool hot_workers_pay(int ranks_on_node) { return effective_cpus() / std::max(1, ranks_on_node) >= 8; }
// out[k] = final_exponentiation(miller_combine(rows + k * N_LINES)), k < count, with the 63 bits of every product cut into `parts` ranges that run
// (recurrence + final exponentiation each) on the host workers; the caller's thread takes one task itself.  The final exponentiation is a
// homomorphism into the cyclotomic subgroup, so  value = conj?( ((E_1^(2^n_2) E_2)^(2^n_3) E_3) ... )  with Granger-Scott squarings: the same
// field element, ~0.57 ms instead of ~0.74 ms per product on the critical path of EVERY round (3 ranges).
double now_ms(); bool trace_on();
void pairing_values(const Fp12* rows, int count, Fp12* out, int parts = 0) {
    if (parts <= 0) parts = host_pool().size() >= 11 ? (count >= 2 ? 5 : 8) : (count >= 2 ? 3 : 4);      // as many ranges as the workers take in one go
    struct Seg { int k, hi, lo; };
    std::vector<Seg> segs;
    for (int k = 0; k < count; ++k) for (int g = 0; g < parts; ++g) segs.push_back({k, 62 - (63 * g) / parts, 62 - (63 * (g + 1)) / parts + 1});
    std::vector<Fp12> E(segs.size());
    auto work = [&rows, &segs, &E](size_t t) { const Seg& sg = segs[t]; E[t] = final_exponentiation(miller_combine_range(rows + (size_t)sg.k * N_LINES, sg.hi, sg.lo)); };
    const double tp0 = now_ms();
    host_pool().parallel((int)segs.size(), [&work](int t) { work((size_t)t); });
    const double tp1 = now_ms();
    host_pool().parallel(count, [&](int k) {
        Fp12 c = E[(size_t)k * parts];
        for (int g = 1; g < parts; ++g) { const Seg& sg = segs[(size_t)k * parts + g]; for (int b = sg.hi; b >= sg.lo; --b) c = cyclotomic_sqr(c); c = mul(c, E[(size_t)k * parts + g]); }
        out[k] = BLS_X_NEG ? conj(c) : c; });
    if (trace_on()) fprintf(stderr, "[ripp] pairing_values(%d x %d): ranges %.3f ms, join %.3f ms\n", count, parts, tp1 - tp0, now_ms() - tp1);
}
struct Engine;
Engine* g_engine = nullptr;
std::atomic<int> g_live_handles{0};       // ripp_sipp_job / ripp_srs objects holding device memory of the current engine

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_err(std::string(#expr) + ": " + hipGetErrorString(e_)); return RIPP_ERR_DEVICE; } } while (0)
void set_err(const std::string& s) { g_err = s; }
bool trace_on();
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline unsigned nblk(size_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }
inline uint32_t pow2_floor(uint32_t v) { uint32_t p = 1; while ((p << 1) <= v && (p << 1) != 0) p <<= 1; return p; }

// A grow-only device buffer
std::atomic<size_t> g_dev_bytes{0};      // device memory this library holds through DevBuf (scratch, tables, jobs, SRS / vector handles): what ripp_config.mem_cap_bytes bounds
// Out of device memory: before a reserve() gives up, the engine of the calling thread frees its IDLE scratch -- the grow-only line buffer, fold tables and
// partial-product buffers that no reserve() of the CURRENT C-ABI call has touched (Engine::evict_idle) -- and the allocation is tried once more.  Found by the
// uncapped n = 2^24 proof (tools/sipp_2p24.py): the prover legitimately fills the device (281 of 288 GB held), and the verifier called next failed in hipMalloc.
std::atomic<uint64_t> g_call_epoch{1};                  // one tick per C-ABI call that takes the engine (get_engine)
struct Engine;
thread_local Engine* t_engine = nullptr;                // the engine this thread drives (get_engine; the threads of the auxiliary / peer engines set their own)
size_t evict_idle_scratch(Engine* e);
struct DevBuf {
    void* p = nullptr; size_t cap = 0; uint64_t epoch = 0;      // epoch: the call that last reserved this buffer
    int32_t reserve(size_t bytes) {
        epoch = g_call_epoch.load(std::memory_order_relaxed);
        if (bytes <= cap) return RIPP_OK;
        release();
        hipError_t err = hipMalloc(&p, bytes);
        if (err == hipErrorOutOfMemory && t_engine) {
            (void)hipGetLastError();
            const size_t freed = evict_idle_scratch(t_engine);
            if (trace_on()) fprintf(stderr, "[ripp] out of device memory for %zu bytes: %zu bytes of idle scratch freed, retrying\n", bytes, freed);
            err = freed ? hipMalloc(&p, bytes) : err;
        }
        if (err != hipSuccess) { p = nullptr; (void)hipGetLastError(); set_err(std::string("hipMalloc(&p, bytes): ") + hipGetErrorString(err) + " (" + std::to_string(bytes) + " bytes wanted, " + std::to_string(g_dev_bytes.load()) + " held by the library)"); return RIPP_ERR_DEVICE; }
        cap = bytes; g_dev_bytes.fetch_add(bytes, std::memory_order_relaxed); return RIPP_OK;
    }
    void release() { if (p) { (void)hipFree(p); g_dev_bytes.fetch_sub(cap, std::memory_order_relaxed); } p = nullptr; cap = 0; }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

// Engine-owned pinned host staging.  Host data this library PRODUCES (scalar vectors of the verifiers and KZG openings, gathered tails)
// reaches the device through these, never by a hipMemcpy from a short-lived malloc'ed vector: the runtime registers such memory with the
// driver for the copy, and in some processes its release (free -> munmap / heap trim) stalled the NEXT device operation by 15-40 ms, in
// steps of 10 ms (measured on the n = 2^13..2^14 SIPP verifier: 13.7 ms became 30-50 ms; tools/kdev/verify_lat2.py).
struct PinBuf {
    void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false;
    bool blocking = false;               // set before reserve(): waits SLEEP instead of spinning (buffers that are only waited for while the statement hash runs)
    bool ev_blocking = false;            // the flag `ev` was created with
    int32_t wait() { if (pending) { HIPCHK(hipEventSynchronize(ev)); pending = false; } return RIPP_OK; }      // the previous copy out of this buffer has landed
    int32_t reserve(size_t bytes) {
        int32_t rc = wait(); if (rc) return rc;
        if (ev && ev_blocking != blocking) { (void)hipEventDestroy(ev); ev = nullptr; }      // (a parked buffer adopted by a call that waits differently)
        if (!ev) { HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | (blocking ? hipEventBlockingSync : 0))); ev_blocking = blocking; }
        if (bytes <= cap) return RIPP_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        HIPCHK(hipHostMalloc(&p, bytes, hipHostMallocDefault)); cap = bytes; return RIPP_OK;
    }
    // dst <- this buffer's first `bytes` on stream st (the caller filled it after reserve())
    int32_t send(void* dst, size_t bytes, hipStream_t st) {
        HIPCHK(hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, st)); HIPCHK(hipEventRecord(ev, st)); pending = true; return RIPP_OK;
    }
    void release() { if (pending && ev) (void)hipEventSynchronize(ev); pending = false; if (p) (void)hipHostFree(p); p = nullptr; cap = 0; if (ev) (void)hipEventDestroy(ev); ev = nullptr; }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

// scratch of ONE in-flight MSM (two MSMs of a GIPA-with-SSM round run side by side on two streams)
struct MsmScratch {
    DevBuf digits, hist, offs, cursor, slotoffs, spw, sorted, slots, buckets, seg, seg2, win, out;
    DevBuf ext;                          // bases and their endomorphism images in the carry-free form (fq_msm.hpp)
    DevBuf flags;                        // one byte per slot: exceptional additions, redone by k_msm_slot_sum_fix[_vm]
    DevBuf nflag;                        // RIPP_TRACE: number of flagged slots of the last MSM
    void* host_out = nullptr;            // pinned landing zone for the result
    void release() { for (DevBuf* b : {&digits, &hist, &offs, &cursor, &slotoffs, &spw, &sorted, &slots, &buckets, &seg, &seg2, &win, &out, &ext, &flags, &nflag}) b->release(); if (host_out) (void)hipHostFree(host_out); host_out = nullptr; }
};

struct Timer {   // HIP-event stopwatch on the engine stream
    hipEvent_t a{}, b{};
    int32_t init() { HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); return RIPP_OK; }
    void destroy() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); a = b = nullptr; }
};

struct Engine {
    int device = -1;
    hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: the G1 half of a fold runs beside the G2 half
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;         // phase stopwatch (scale / fold), reused by every call
    int n_simd = 1024;
    // scratch
    DevBuf lines, partA, partB, jacG1, jacG2, tmpA, tmpB, tmpR, affG1, affG2;
    DevBuf qtab;                          // [u^j]Q table of the GLS G2 fold
    MsmScratch msm_scratch[2];
    DevBuf kzg_q[2];                      // quotient-polynomial coefficients of the (up to two concurrent) KZG openings
    PinBuf stage[4];                      // pinned staging of host-produced vectors: [0], [1] scalar vectors of the two concurrent MSMs, [2] r-powers, [3] gathered tails
    DevBuf kzg_bases[2];                  // sharded openings: this rank's residue class of the SRS powers, gathered contiguously
    hipStream_t stream3 = nullptr;        // second MSM of a pair
    hipStream_t stream4 = nullptr;        // unscaled twin of m_a in the implicit-shift TIPP core
    hipEvent_t ev_join4 = nullptr;
    // Fold tables (kernels.hpp: k_odd_multiples / k_fold_g*_tab) live in the ENGINE, not in jobs or per-call vectors: they are GB-sized at
    // n = 2^20 and a hipMalloc of that size costs ~100 ms, which a one-shot ripp_sipp_prove would pay on every call.  tab_owner names the
    // job whose round-0 tables they hold; a job that finds another owner falls back to the table-free fold.
    DevBuf fold_tab1, fold_mult, fold_tab, fold_jac1, fold_jac2;
    DevBuf fix_flags, scale_flags;        // one byte per lane of a carry-free G2 fold / table / scaling kernel: lanes with an exceptional addition, redone by the *_fix kernel behind it
    const void* tab_owner = nullptr;
    // Device and pinned buffers of the last ONE-SHOT proof (ripp_sipp_prove[_sharded]), adopted by the next one: hipMalloc / hipFree of ~1 GB and four
    // pinned staging buffers per call cost ~10 ms of every host-slice proof -- most of it the frees, AFTER the digest, where nothing hides them
    // (host slices 442.6 ms against 430.3 ms for the same proof on a resident job).  Freed by ripp_release_scratch / ripp_shutdown like the other scratch.
    struct JobBufCache { DevBuf d[15]; PinBuf p[4]; bool full = false; void release() { for (DevBuf& b : d) b.release(); for (PinBuf& b : p) b.release(); full = false; } } job_cache;
    size_t msm_chunk_min = (size_t)1 << 20;                               // host-slice MSMs of >= this many G1 bases (half as many G2 bases: the same bytes) run as two halves on two streams (msm_impl: the second half's upload beside the first half's additions)
    size_t msm_lds_sort_min = 0;                                          // MSMs of >= this many terms (after the GLV / GLS split) sort through LDS tiles (msm.hpp k_msm_hist_lds / k_msm_scatter_lds); the lane-per-term sort below it (A/B)
    const void* g2tab_hi = nullptr; size_t g2tab_half = 0;               // in-round G2 fold tables built ahead of the challenge (fold_g2_table_build): the vector half they were built over
    size_t vm_scale_max = (size_t)1 << 14;                                // per-element G1 scalings of <= this many elements run on the VM (measured: direct product 6.1 -> 3.8 ms at 2^13, 7.1 -> 6.2 ms at 2^14, level at 2^15)
    size_t vm_joint_max = (size_t)1 << 13;                                // folds with <= this many outputs (and more than vm_fold_max) use the joint one-group-per-element VM forms
    size_t lp_fq_min = 0;                 // pairs per launch from which k_line_products_q replaces k_line_products: always (11 % faster per launch: 5.8 vs 6.5 ms at the proof's launch
                                          // mix; it keeps ~47 dwords per lane in scratch, which costs HBM traffic, not time -- the kernel is issue-bound).  RIPP_LP_FQ_MIN=4294967295 / RIPP_NO_FQ select k_line_products
    size_t ml_fq_min = 0;                 // pairs per launch from which k_miller_lines_q (fq_miller.hpp) replaces k_miller_lines: every throughput launch (the VM kernel covers the small ones)
    size_t fq_min_g1 = (size_t)1 << 12;   // ... the G1 NAF fold (one 128-step chain per lane either way): the carry-free kernel wherever the lane-per-element form runs at all
    size_t fq_min = (size_t)1 << 15;      // the carry-free fold kernels (fq_curve.hpp, fq_curve2.hpp): from 2^15 outputs (round 4 of an n = 2^20 proof: 5.4 -> 3.9 ms; below that the 12 x 32-bit forms have the shorter chains)
    size_t fold_tab_min = 32768;          // G2 folds of at least this many elements build in-round odd-multiple tables
    size_t msm_vm_merge_max = 16384;      // buckets (all windows) up to which the bucket merge runs on the field VM
    hipStream_t stream5 = nullptr;        // second G2 fold of a small GIPA round (own scratch there, so it need not queue behind the first)
    hipEvent_t ev_join5 = nullptr;
    Fp12* pinned_rows = nullptr;          // pinned host landing zone for per-step products
    size_t pinned_rows_cap = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_lines, ev_prod;   // per-launch event pairs of the two dominant kernels
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_fold_async;       // folds nobody waited for (pipelined / pre-evaluated rounds): their device time is added to stats.fold_ms when the proof ends
    size_t tail_pipe_max = (size_t)1 << 11;                               // SIPP rounds of at most this length run pipelined (job_tail_enqueue); 0 = off
    size_t vm_tree_max = (size_t)1 << 16;                                 // tree levels with <= this many products use the VM Fp12 multiplier
    size_t vm_fold_max = (size_t)1 << 11;                                 // folds with <= this many outputs use the VM scalar multiplications
    DevBuf vm_flag;
    size_t vm_lines_max = (size_t)1 << 15;                                // launches with <= this many pairs use the 16-lanes-per-pair VM line kernel (measured crossover)
    size_t gls_split_max = (size_t)1 << 14;                               // rounds with <= this many outputs use the 4-lane GLS fold
    size_t max_pairs_per_batch = (size_t)1 << 19;                        // lines buffer cap: 2^19 pairs * 19.6 KB = 10.3 GB (pairs_cap() lowers it when the device is short of memory)
    // ---- memory-aware degradation (ripp_config.mem_cap_bytes) --------------------------------------------------------------
    // The optional structures of a large proof are sized against what the device can still give: the line buffer first (without it nothing runs:
    // pairs_cap halves the batch until it fits), then the round-0 fold tables (13.6 KB per element in the three-quarter form: job_precompute_round0
    // steps down to half-vector tables of four multiples, then to the two pre-doubled bases, then to nothing), the in-round G2 tables
    // (fold_g2_table_pays).  Every tier computes the same group elements.  mem_cap = 0: what hipMemGetInfo reports free, less a margin.
    size_t mem_cap = 0;
    int mem_tier = 0;                     // the deepest fall-back this call took: 0 none, 1 half-vector tables, 2 pre-doubled bases only, 3 no round-0 precomputation; +8: the line buffer was cut; +16: an in-round G2 table fold took the split form
    bool mem_fits(size_t need, size_t held) const {
        if (need <= held) return true;
        const size_t extra = need - held;
        if (mem_cap) return g_dev_bytes.load(std::memory_order_relaxed) + extra <= mem_cap;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) return true;
        return extra + std::max<size_t>((size_t)2 << 30, tot / 50) <= fr;
    }
    // pairs one launch may take through the line buffer (<= want): probes the device only when the buffer would have to grow
    size_t pairs_cap(size_t want) {
        want = std::max<size_t>(1, std::min(want, max_pairs_per_batch));
        const size_t per = (size_t)N_LINES * LINE_CHUNKS * sizeof(uint4);
        auto bytes = [per](size_t pairs) { return ((pairs + 63) & ~(size_t)63) * per + (size_t)MAX_PRODUCTS * 64 * per; };      // (+ the stride rounding of up to MAX_PRODUCTS products)
        if (bytes(want) <= lines.cap) return want;
        size_t w = want;
        while (w > 4096 && !mem_fits(bytes(w), lines.cap)) w = (w + 1) / 2;
        if (w < want) mem_tier |= 8;
        return w;
    }
    ripp_stats stats{};
    // A second set of streams and scratch on the SAME device: two independent latency-bound provers of one call (aggregate_proofs' TIPP and
    // TIPAWithSSM sub-proofs) run side by side, each driven by its own host thread, instead of taking turns waiting for host and device.
    // In-process multi-device dispatch of the stateless trait calls (ripp_config.n_devices): one more Engine per extra device, each driven by its own host
    // thread for the duration of one call (device_slots / run_on_devices below).  RIPP_VIRTUAL_DEVICES=G (one-GPU test rig): G slots, all on the bound device.
    std::vector<Engine*> peers; bool virtual_devices = false;
    Engine* aux = nullptr;
    Engine* aux_engine() {
        if (!aux) { Engine* a = new Engine(); if (a->init(device) != RIPP_OK) { delete a; return nullptr; } aux = a; }
        aux->refresh_switches(); aux->stats = ripp_stats{};
        return aux;
    }
    // run-time switches: read from the environment ONCE per C-ABI call (get_engine), never inside round loops
    struct Switches { bool no_vm = false, no_precompute = false, no_fold_tables = false, no_msm_glv = false, lp_one_lane = false, no_endo = false, no_fq = false, no_xscale = false, no_share = false, no_fuse = false, no_prebuild = false, fuse_tables = false, no_job_cache = false, no_lp_kara = false; } sw;
    // crossover sizes: the member initialisers above are the defaults, the environment overrides them PER CALL (a test or
    // an A/B run flips them on a live engine)
    struct Sizes { size_t vm_lines_max, vm_fold_max, vm_tree_max, gls_split_max, msm_vm_merge_max, fold_tab_min, fq_min, lp_fq_min, vm_joint_max, vm_scale_max, tail_pipe_max, ml_fq_min, fq_min_g1, msm_lds_sort_min, msm_chunk_min; } defaults{};
    MsmTune msm_tune;
    // the hash-window look-ahead plan and a few whole-call choices (ripp_config: look_eighths, ranks_per_device, look_static, quiet_waits, agg_sequential, scale_no_fq)
    double cal_ms_per_pair = 0, cal_hash_bytes_per_ms = 0;        // look_plan's rates as measured by the last large proof of this process (0: not yet)
    int look_eighths = -1; double ranks_per_device = 1.0; bool look_static = false, quiet_waits_cfg = false, agg_sequential = false, scale_no_fq = false;
    int hot_workers_cfg = 0;               // 0 automatic, 1 always, 2 never (ripp_config.hot_workers / RIPP_HOT_WORKERS)
    uint32_t comm_timeout_ms = 0, plan_derate_pct = 0, n_devices_cfg = 0;      // ripp_config members of the same names (RIPP_COMM_TIMEOUT_MS, RIPP_PLAN_DERATE_PCT, RIPP_N_DEVICES)
    // Precedence: built-in defaults < ripp_configure() < environment variables (a debug / A-B override).  This function is the ONLY place of the
    // library that reads RIPP_* configuration from the environment (RIPP_TRACE aside), once per C-ABI call (get_engine) -- never inside a proof.
    void refresh_switches() {
        msm_tune = MsmTune();
        vm_lines_max = defaults.vm_lines_max; vm_fold_max = defaults.vm_fold_max; vm_tree_max = defaults.vm_tree_max; gls_split_max = defaults.gls_split_max;
        msm_vm_merge_max = defaults.msm_vm_merge_max; fold_tab_min = defaults.fold_tab_min; fq_min = defaults.fq_min; lp_fq_min = defaults.lp_fq_min; vm_joint_max = defaults.vm_joint_max;
        vm_scale_max = defaults.vm_scale_max; tail_pipe_max = defaults.tail_pipe_max; ml_fq_min = defaults.ml_fq_min; fq_min_g1 = defaults.fq_min_g1; msm_lds_sort_min = defaults.msm_lds_sort_min; msm_chunk_min = defaults.msm_chunk_min;
        sw = Switches(); look_eighths = -1; ranks_per_device = 1.0; look_static = quiet_waits_cfg = agg_sequential = scale_no_fq = false; mem_cap = 0; hot_workers_cfg = 0; comm_timeout_ms = plan_derate_pct = n_devices_cfg = 0;
        if (g_cfg_set) {
            const ripp_config& c = g_cfg;
            sw.no_vm = c.no_vm; sw.no_precompute = c.no_precompute; sw.no_fold_tables = c.no_fold_tables; sw.no_msm_glv = c.no_msm_glv; sw.lp_one_lane = c.lp_one_lane;
            sw.no_endo = c.no_endo; sw.no_fq = c.no_fq; sw.no_xscale = c.no_xscale; sw.no_share = c.no_share; sw.no_fuse = c.no_fuse; sw.fuse_tables = c.fuse_tables; scale_no_fq = c.scale_no_fq; agg_sequential = c.agg_sequential; look_static = c.look_static; quiet_waits_cfg = c.quiet_waits;
            look_eighths = c.look_eighths; ranks_per_device = c.ranks_per_device > 1 ? (double)c.ranks_per_device : 1.0;
            msm_tune.c = c.msm_c; msm_tune.ch = c.msm_ch; msm_tune.gmin = c.msm_gmin; sw.no_prebuild = c.no_prebuild;
            mem_cap = (size_t)c.mem_cap_bytes; hot_workers_cfg = (int)c.hot_workers; sw.no_job_cache = c.no_job_cache != 0;
            sw.no_lp_kara = c.no_lp_karatsuba != 0; comm_timeout_ms = c.comm_timeout_ms; plan_derate_pct = c.plan_derate_pct; n_devices_cfg = c.n_devices;
            vm_lines_max = c.vm_lines_max; vm_fold_max = c.vm_fold_max; vm_tree_max = c.vm_tree_max; gls_split_max = c.gls_split_max; msm_vm_merge_max = c.msm_vm_merge_max; fold_tab_min = c.fold_tab_min;
            fq_min = c.fq_min; lp_fq_min = c.lp_fq_min; vm_joint_max = c.vm_joint_max; vm_scale_max = c.vm_scale_max; tail_pipe_max = c.tail_pipe_max; ml_fq_min = c.ml_fq_min; fq_min_g1 = c.fq_min_g1; msm_lds_sort_min = c.msm_lds_sort_min; msm_chunk_min = c.msm_chunk_min;
        }
        { const char* s; if ((s = std::getenv("RIPP_MSM_C"))) msm_tune.c = std::atoi(s); if ((s = std::getenv("RIPP_MSM_CH"))) msm_tune.ch = (uint32_t)std::strtoul(s, nullptr, 10); if ((s = std::getenv("RIPP_MSM_GMIN"))) msm_tune.gmin = (uint32_t)std::strtoul(s, nullptr, 10); }
        auto env_sz = [](const char* k, size_t& v) { if (const char* s = std::getenv(k)) v = (size_t)std::strtoull(s, nullptr, 10); };
        env_sz("RIPP_VM_LINES_MAX", vm_lines_max); env_sz("RIPP_VM_FOLD_MAX", vm_fold_max); env_sz("RIPP_VM_TREE_MAX", vm_tree_max);
        env_sz("RIPP_GLS_SPLIT_MAX", gls_split_max); env_sz("RIPP_MSM_VM_MERGE_MAX", msm_vm_merge_max); env_sz("RIPP_FOLD_TAB_MIN", fold_tab_min);
        env_sz("RIPP_FQ_MIN", fq_min); env_sz("RIPP_LP_FQ_MIN", lp_fq_min); env_sz("RIPP_VM_JOINT_MAX", vm_joint_max);
        env_sz("RIPP_MSM_LDS_SORT_MIN", msm_lds_sort_min); env_sz("RIPP_MSM_CHUNK_MIN", msm_chunk_min);
        env_sz("RIPP_VM_SCALE_MAX", vm_scale_max); env_sz("RIPP_TAIL_PIPE_MAX", tail_pipe_max); env_sz("RIPP_ML_FQ_MIN", ml_fq_min); env_sz("RIPP_FQ_MIN_G1", fq_min_g1);
        auto env_on = [](const char* k, bool& v) { if (std::getenv(k)) v = true; };
        env_on("RIPP_NO_VM", sw.no_vm); env_on("RIPP_NO_PRECOMPUTE", sw.no_precompute); env_on("RIPP_NO_FOLD_TABLES", sw.no_fold_tables); env_on("RIPP_NO_MSM_GLV", sw.no_msm_glv);
        env_on("RIPP_LP_ONE_LANE", sw.lp_one_lane);
        env_on("RIPP_NO_ENDO", sw.no_endo);            // plain scalar multiplications in the folds / scaling (no GLV, no psi)
        env_on("RIPP_NO_XSCALE", sw.no_xscale);        // G2 folds always on the plain vector with the full-width x^-1
        env_on("RIPP_NO_LP_KARA", sw.no_lp_kara);      // stage 2 of the pairing product with the six-product sums of k_line_products_q instead of the Karatsuba form (BLS12-381)
        env_on("RIPP_NO_FQ", sw.no_fq);                // the 12 x 32-bit forms of the kernels that have a carry-free twin (fq_curve.hpp)
        env_on("RIPP_FUSE_TABLES", sw.fuse_tables);    // build the three-quarter tables whatever the look-ahead plan (tests: round 0 then folds ALONE over them when x1 is late)
        env_on("RIPP_NO_JOB_CACHE", sw.no_job_cache);  // one-shot proofs allocate and free their job buffers per call (ripp_config.no_job_cache)
        env_sz("RIPP_MEM_CAP_BYTES", mem_cap);         // device memory the library may hold (ripp_config.mem_cap_bytes; 0 = automatic)
        if (const char* s = std::getenv("RIPP_HOT_WORKERS")) hot_workers_cfg = std::atoi(s) ? 1 : 2;
        { auto env_u32 = [](const char* k, uint32_t& v) { if (const char* s = std::getenv(k)) v = (uint32_t)std::strtoul(s, nullptr, 10); };
          env_u32("RIPP_COMM_TIMEOUT_MS", comm_timeout_ms); env_u32("RIPP_PLAN_DERATE_PCT", plan_derate_pct); env_u32("RIPP_N_DEVICES", n_devices_cfg);
          virtual_devices = false; if (const char* s = std::getenv("RIPP_VIRTUAL_DEVICES")) { n_devices_cfg = (uint32_t)std::strtoul(s, nullptr, 10); virtual_devices = true; } }
        env_on("RIPP_NO_PREBUILD", sw.no_prebuild);    // in-round G2 fold tables after the challenge (fold_g2_table), not in the host phase before it (job_prebuild_g2_tables)
        env_on("RIPP_NO_FUSE", sw.no_fuse);            // rounds 0 and 1 always fold one after the other (no three-quarter tables, no job_fold_fused)
        env_on("RIPP_NO_SHARE", sw.no_share);          // every pairing product walks its own G2 chain (no ChainSets grouping, no merged round 0 + look-ahead)
        env_on("RIPP_SCALE_NO_FQ", scale_no_fq); env_on("RIPP_AGG_SEQUENTIAL", agg_sequential); env_on("RIPP_LOOK_STATIC", look_static); env_on("RIPP_QUIET_WAITS", quiet_waits_cfg);
        if (const char* s = std::getenv("RIPP_LOOK_ITEMS")) look_eighths = 8 * std::max(0, std::atoi(s));            // whole (round, side) items
        if (const char* s = std::getenv("RIPP_LOOK_EIGHTHS")) look_eighths = std::max(0, std::atoi(s));               // 8 k + f: k items and f/8 of the next
        if (const char* s = std::getenv("RIPP_RANKS_PER_DEVICE")) ranks_per_device = std::max(1.0, std::atof(s));
        // (both builds run the carry-free throughput kernels: fq_curve2.hpp FQ2_BETA, fq_miller.hpp / fq_line_products.hpp per twist type)
#if !defined(RIPP_AB_KERNELS) || defined(RIPP_BLS12_377)
        sw.lp_one_lane = false;                        // k_line_products1 is compiled into A/B builds of the BLS12-381 library only (-DRIPP_AB_KERNELS)
#endif
    }

    int32_t init(int dev) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_err("no HIP device available (libripp_hip has no CPU fallback)"); return RIPP_ERR_DEVICE; }
        if (dev < 0 || dev >= n) { set_err("device ordinal out of range"); return RIPP_ERR_ARG; }
        HIPCHK(hipSetDevice(dev));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&stream3, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&stream4, hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&ev_join4, hipEventDisableTiming));
        HIPCHK(hipStreamCreateWithFlags(&stream5, hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&ev_join5, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ev_join3, hipEventDisableTiming));
        HIPCHK(hipEventCreate(&ev_t0)); HIPCHK(hipEventCreate(&ev_t1));
        { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess) n_simd = pr.multiProcessorCount * 4; }
        defaults = Sizes{vm_lines_max, vm_fold_max, vm_tree_max, gls_split_max, msm_vm_merge_max, fold_tab_min, fq_min, lp_fq_min, vm_joint_max, vm_scale_max, tail_pipe_max, ml_fq_min, fq_min_g1, msm_lds_sort_min, msm_chunk_min};
        refresh_switches();
        device = dev;
        return RIPP_OK;
    }
    void destroy() {
        if (aux) { aux->destroy(); delete aux; aux = nullptr; }
        for (Engine* p : peers) { (void)hipSetDevice(p->device); p->destroy(); delete p; }
        if (!peers.empty()) { peers.clear(); (void)hipSetDevice(device); }
        for (DevBuf* b : {&lines, &partA, &partB, &jacG1, &jacG2, &tmpA, &tmpB, &tmpR, &affG1, &affG2, &qtab, &vm_flag, &scale_tab, &fold_tab1, &fold_mult, &fold_tab, &fold_jac1, &fold_jac2, &fix_flags, &scale_flags}) b->release();
        msm_scratch[0].release(); msm_scratch[1].release(); kzg_q[0].release(); kzg_q[1].release(); kzg_bases[0].release(); kzg_bases[1].release(); job_cache.release();
        if (stream3) (void)hipStreamDestroy(stream3);
        if (stream4) (void)hipStreamDestroy(stream4); if (ev_join4) (void)hipEventDestroy(ev_join4);
        if (stream5) (void)hipStreamDestroy(stream5); if (ev_join5) (void)hipEventDestroy(ev_join5);
        if (pinned_rows) (void)hipHostFree(pinned_rows);
        for (PinBuf& b : stage) b.release();
        for (auto& e : ev_lines) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto& e : ev_prod) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto& e : ev_fold_async) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        if (stream) (void)hipStreamDestroy(stream);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (ev_fork) (void)hipEventDestroy(ev_fork); if (ev_join) (void)hipEventDestroy(ev_join); if (ev_join3) (void)hipEventDestroy(ev_join3);
        if (ev_t0) (void)hipEventDestroy(ev_t0); if (ev_t1) (void)hipEventDestroy(ev_t1);
        if (ev_quiet) (void)hipEventDestroy(ev_quiet);
    }
    int32_t ensure_pinned_rows(size_t rows) {
        if (rows <= pinned_rows_cap) return RIPP_OK;
        if (pinned_rows) (void)hipHostFree(pinned_rows);
        HIPCHK(hipHostMalloc((void**)&pinned_rows, rows * sizeof(Fp12), hipHostMallocDefault));
        pinned_rows_cap = rows; return RIPP_OK;
    }
    // quiet_waits (RIPP_QUIET_WAITS, an experiment kept for A/B): stream waits go through an event created with hipEventBlockingSync, so the
    // prover's thread sleeps instead of spinning next to the hashing core while the statement hash runs.
    bool quiet_waits = false; hipEvent_t ev_quiet = nullptr;
    int32_t sync() {
        if (quiet_waits) {
            if (!ev_quiet) HIPCHK(hipEventCreateWithFlags(&ev_quiet, hipEventDisableTiming | hipEventBlockingSync));
            HIPCHK(hipEventRecord(ev_quiet, stream)); HIPCHK(hipEventSynchronize(ev_quiet)); return RIPP_OK;
        }
        HIPCHK(hipStreamSynchronize(stream)); return RIPP_OK;
    }

    // ---- event bookkeeping for the roofline figures -----------------------------------------------------
    int32_t mark(std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, bool begin) {
        if (begin) { hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); v.emplace_back(a, b); HIPCHK(hipEventRecord(a, stream)); }
        else HIPCHK(hipEventRecord(v.back().second, stream));
        return RIPP_OK;
    }
    void collect_kernel_stats() {   // call after a stream sync
        auto drain = [](std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, double& sum, uint64_t& cnt) {
            for (auto& e : v) { float ms = 0; if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { sum += ms; ++cnt; } (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
            v.clear(); };
        drain(ev_lines, stats.kernel_miller_lines_ms_sum, stats.kernel_miller_lines_launches);
        drain(ev_prod, stats.kernel_line_products_ms_sum, stats.kernel_line_products_launches);
        uint64_t nf = 0; drain(ev_fold_async, stats.fold_ms, nf);
    }

    // ---- per-element G1 scalar multiplication (GLV + signed windows, scale.hpp); out may not alias base -----------------
    DevBuf scale_tab;
    int32_t scale_g1_dev(const G1A* base, uint32_t base_stride, const Fr* k, size_t n, G1J* out, hipStream_t st = nullptr) {
        if (n == 0) return RIPP_OK;
        if (!st) st = stream;
        if (sw.no_endo) {                                     // 255-bit double-and-add (the reference's `a.mul(r)`, sipp/src/lib.rs:61-65)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scale_pts<Fp>), dim3(nblk(n, 256)), dim3(256), 0, st, base, base_stride, k, (uint32_t)n, out);
            HIPCHK(hipGetLastError());
            return RIPP_OK;
        }
        if (n <= vm_scale_max && !sw.no_vm) {                  // few elements: one VM group per element instead of a lone lane (~1.3 ms instead of ~3.9 ms)
            hipLaunchKernelGGL(k_vm_scale_g1, dim3(nblk(n, 4 * VM_EPW)), dim3(256), 4 * VM_EPW * VM_SCALE_SLOTS * sizeof(VmSlot), st, base, base_stride, k, (uint32_t)n, out);
            HIPCHK(hipGetLastError());
            return RIPP_OK;
        }
        int32_t rc = scale_tab.reserve((size_t)SCALE_TAB * G1J_CHUNKS * n * sizeof(uint4)); if (rc) return rc;
        if (!sw.no_fq && !scale_no_fq)      // the carry-free twin (fq_scale.hpp)
        {
            if ((rc = scale_flags.reserve(n + 16))) return rc;
            hipLaunchKernelGGL(k_scale_g1_glv_q, dim3(nblk(n, 256)), dim3(256), 0, st, base, base_stride, k, (uint32_t)n, scale_tab.as<uint4>(), out, scale_flags.as<uint8_t>());
            hipLaunchKernelGGL(k_scale_g1_fix, dim3(FIX_GRID), dim3(64), 0, st, base, base_stride, k, (uint32_t)n, out, scale_flags.as<uint8_t>());
        }
        else
        hipLaunchKernelGGL(k_scale_g1_glv, dim3(nblk(n, 256)), dim3(256), 0, st, base, base_stride, k, (uint32_t)n, scale_tab.as<uint4>(), out);
        HIPCHK(hipGetLastError());
        return RIPP_OK;
    }

    // ---- normalisation (device in, device out) ------------------------------------------------------------
    template <class F> int32_t normalize_dev(const Jac<F>* in, size_t n, Affine<F>* out, hipStream_t st = nullptr) {
        if (n == 0) return RIPP_OK;
        if (!st) st = stream;
        // one inversion (~600 Fp products) per lane: amortise over up to 16 points but keep >= ~64K lanes busy
        uint32_t K = (uint32_t)std::min<size_t>(16, std::max<size_t>(1, n / 65536));
        uint32_t T = (uint32_t)((n + K - 1) / K);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_normalize<F>), dim3(nblk(T, 256)), dim3(256), 0, st, in, (uint32_t)n, out, T);
        HIPCHK(hipGetLastError());
        return RIPP_OK;
    }


    // ---- Pippenger MSM over device-resident affine bases and Montgomery scalars; result (Jacobian) to host --------
    // msm_launch enqueues the whole pipeline on `st` with scratch `ms` and leaves the result in ms.host_out (valid after a
    // sync of `st`); msm_dev is the synchronous single-MSM form.
    // bases_arrive (optional): called once the digit sort -- which reads only the scalars -- is enqueued; it brings the bases to the device on another
    // stream and makes `st` wait for them, so that a host-slice MSM uploads its bases beside the sort instead of in front of it.
    template <class F> int32_t msm_launch(MsmScratch& ms, hipStream_t st, const Affine<F>* bases, const Fr* scalars, size_t n, const std::function<int32_t()>* bases_arrive = nullptr) {
        if (!ms.host_out) HIPCHK(hipHostMalloc(&ms.host_out, sizeof(G2J), hipHostMallocDefault));
        if (n == 0) { *reinterpret_cast<Jac<F>*>(ms.host_out) = jac_inf<F>(); return bases_arrive ? (*bases_arrive)() : RIPP_OK; }
        const bool no_glv = sw.no_msm_glv;
        const size_t nreal = n;
        const MsmPlan p = msm_plan(nreal, no_glv ? 1 : std::is_same<F, Fp>::value ? 2 : 4, msm_tune);
        n = p.n;                                                                  // terms (2 * nreal in the GLV form, 4 * nreal in the GLS form)
        const size_t nwb = (size_t)p.nwin * p.nb;
        const uint32_t max_slots = (uint32_t)(n / p.ch + std::min<size_t>(p.nb, n) + 1);
        uint32_t nseg = (p.nb + p.seg - 1) / p.seg;
        int32_t rc;
        if ((rc = ms.digits.reserve((size_t)p.nwin * n * sizeof(uint16_t))) || (rc = ms.hist.reserve(nwb * 4)) || (rc = ms.offs.reserve(nwb * 4)) ||
            (rc = ms.cursor.reserve(nwb * 4)) || (rc = ms.slotoffs.reserve(nwb * 4)) || (rc = ms.spw.reserve(p.nwin * 4)) ||
            (rc = ms.sorted.reserve((size_t)p.nwin * n * 4)) || (rc = ms.slots.reserve((size_t)p.nwin * max_slots * sizeof(Jac<F>))) ||
            (rc = ms.buckets.reserve(nwb * sizeof(Jac<F>))) || (rc = ms.seg.reserve((size_t)p.nwin * nseg * sizeof(Jac<F>))) ||
            (rc = ms.seg2.reserve((size_t)p.nwin * ((nseg + MSM_SEG_FAN - 1) / MSM_SEG_FAN) * sizeof(Jac<F>))) ||
            (rc = ms.win.reserve(64 * sizeof(Jac<F>))) || (rc = ms.out.reserve(sizeof(Jac<F>)))) return rc;
        HIPCHK(hipMemsetAsync(ms.hist.p, 0, nwb * 4, st));
        const bool lds_sort = n >= msm_lds_sort_min && p.nb <= 8192;   // count and rank through LDS tiles (msm.hpp; measured no slower at any size, 2.3 ms faster at n = 2^20)
        const uint32_t tile = msm_sort_tile(p);
        hipLaunchKernelGGL(k_msm_digits, dim3(nblk(nreal, 256)), dim3(256), 0, st, scalars, p, ms.digits.as<uint16_t>(), lds_sort ? nullptr : ms.hist.as<uint32_t>());
        if (lds_sort) hipLaunchKernelGGL(k_msm_hist_lds, dim3(nblk(n, tile), p.nwin), dim3(MSM_SORT_BLOCK), 0, st, ms.digits.as<uint16_t>(), p, tile, ms.hist.as<uint32_t>());
        hipLaunchKernelGGL(k_msm_scan, dim3(p.nwin), dim3(1024), 0, st, ms.hist.as<uint32_t>(), p, ms.offs.as<uint32_t>(), ms.cursor.as<uint32_t>(), ms.slotoffs.as<uint32_t>(), ms.spw.as<uint32_t>());
        if (lds_sort) hipLaunchKernelGGL(k_msm_scatter_lds, dim3(nblk(n, tile), p.nwin), dim3(MSM_SORT_BLOCK), 0, st, ms.digits.as<uint16_t>(), p, tile, ms.cursor.as<uint32_t>(), ms.sorted.as<uint32_t>());
        else hipLaunchKernelGGL(k_msm_scatter, dim3(nblk(n, 256)), dim3(256), 0, st, ms.digits.as<uint16_t>(), p, ms.cursor.as<uint32_t>(), ms.sorted.as<uint32_t>());
        if (bases_arrive && (rc = (*bases_arrive)())) return rc;
        // RIPP_NO_VM keeps every stage on single lanes in Jacobian coordinates (the A/B and fallback form); otherwise the stages after
        // the gather work on homogeneous coordinates and the ones with few points run on the field VM (msm.hpp)
        const bool hom = !sw.no_vm;
        const size_t vm_lds = 4 * VM_EPW * VmCurve<F>::SLOTS * sizeof(VmSlot);
        if (!sw.no_fq) {        // gathered additions on the carry-free form over the extended base array (fq_msm.hpp), both curves
            const int split = (int)(n / nreal);
            if ((rc = ms.ext.reserve(n * sizeof(Affine<F>))) || (rc = ms.flags.reserve((size_t)p.nwin * max_slots + 16))) return rc;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_extend_q<F>), dim3(nblk(nreal, 256), split), dim3(256), 0, st, bases, (uint32_t)nreal, split, ms.ext.as<QAff<F>>());
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_slot_sum_q<F>), dim3(nblk(max_slots, 64), p.nwin), dim3(64), 0, st, ms.ext.as<QAff<F>>(), p, ms.hist.as<uint32_t>(), ms.offs.as<uint32_t>(),
                               ms.slotoffs.as<uint32_t>(), ms.spw.as<uint32_t>(), ms.sorted.as<uint32_t>(), ms.slots.as<Jac<F>>(), max_slots, hom, ms.flags.as<uint8_t>());
            if (hom) {      // flagged slots on the field VM, one wave each (fq_msm.hpp)
                if ((rc = ms.nflag.reserve(16))) return rc;
                if (trace_on()) HIPCHK(hipMemsetAsync(ms.nflag.p, 0, 4, st));
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_slot_sum_fix_vm<F>), dim3(FIX_GRID), dim3(64), VM_EPW * VmCurve<F>::SLOTS * sizeof(VmSlot), st, bases, p, ms.hist.as<uint32_t>(), ms.offs.as<uint32_t>(),
                                   ms.slotoffs.as<uint32_t>(), ms.spw.as<uint32_t>(), ms.sorted.as<uint32_t>(), ms.slots.as<Jac<F>>(), max_slots, ms.flags.as<uint8_t>(), trace_on() ? ms.nflag.as<uint32_t>() : nullptr);
                if (trace_on()) {
                    uint32_t nf = 0; HIPCHK(hipMemcpyAsync(&nf, ms.nflag.p, 4, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
                    fprintf(stderr, "[ripp] msm %s n=%zu: %u of %u slots had an exceptional addition (redone on the field VM)\n", std::is_same<F, Fp>::value ? "G1" : "G2", nreal, nf, (uint32_t)p.nwin * max_slots);
                }
            } else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_slot_sum_fix<F>), dim3(FIX_GRID), dim3(64), 0, st, bases, p, ms.hist.as<uint32_t>(), ms.offs.as<uint32_t>(),
                               ms.slotoffs.as<uint32_t>(), ms.spw.as<uint32_t>(), ms.sorted.as<uint32_t>(), ms.slots.as<Jac<F>>(), max_slots, hom, ms.flags.as<uint8_t>());
        } else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_slot_sum<F>), dim3(nblk(max_slots, 64), p.nwin), dim3(64), 0, st, bases, p, ms.hist.as<uint32_t>(), ms.offs.as<uint32_t>(),
                           ms.slotoffs.as<uint32_t>(), ms.spw.as<uint32_t>(), ms.sorted.as<uint32_t>(), ms.slots.as<Jac<F>>(), max_slots, hom);
        uint32_t passes = 0;                                                       // a bucket holds at most n / ch + 1 slots
        for (uint32_t stride = 1; passes < (uint32_t)MSM_GROUP_PASSES && n / p.ch + 1 > (size_t)p.gmin * stride; stride *= MSM_SLOT_GROUP, ++passes)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_slot_group<F>), dim3(nblk(max_slots, 64), p.nwin), dim3(64), 0, st, p, ms.hist.as<uint32_t>(),
                               ms.slotoffs.as<uint32_t>(), ms.spw.as<uint32_t>(), ms.slots.as<Jac<F>>(), max_slots, stride, hom);
        if (hom && nwb <= msm_vm_merge_max)                                        // few buckets: 16 lanes per bucket still fit the chip
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_vm_merge<F>), dim3(nblk(p.nb, 4 * VM_EPW), p.nwin), dim3(256), vm_lds, st, p, ms.hist.as<uint32_t>(), ms.slotoffs.as<uint32_t>(),
                               ms.slots.as<Jac<F>>(), max_slots, ms.buckets.as<Jac<F>>(), passes);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_bucket_merge<F>), dim3(nblk(p.nb, 64), p.nwin), dim3(64), 0, st, p, ms.hist.as<uint32_t>(), ms.slotoffs.as<uint32_t>(),
                               ms.slots.as<Jac<F>>(), max_slots, ms.buckets.as<Jac<F>>(), passes, hom);
        Jac<F>* cur = ms.seg.as<Jac<F>>(); Jac<F>* nxt = ms.seg2.as<Jac<F>>();
        if (hom) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_vm_segments<F>), dim3(nblk(nseg, 4 * VM_EPW), p.nwin), dim3(256), vm_lds, st, p, ms.buckets.as<Jac<F>>(), cur, nseg);
            while (nseg > 1) {                                                     // 4-ary tree down to one sum per window
                const uint32_t nout = (nseg + MSM_SEG_FAN - 1) / MSM_SEG_FAN;
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_vm_reduce<F>), dim3(nblk(nout, 4 * VM_EPW), p.nwin), dim3(256), vm_lds, st, cur, nseg, nxt, nout);
                std::swap(cur, nxt); nseg = nout;
            }
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_finish_vm<F>), dim3(1), dim3(64), VM_EPW * VmCurve<F>::SLOTS * sizeof(VmSlot), st, p, cur, ms.win.as<Jac<F>>(), ms.out.as<Jac<F>>());
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_segments<F>), dim3(nblk(nseg, 64), p.nwin), dim3(64), 0, st, p, ms.buckets.as<Jac<F>>(), cur, nseg);
            while (nseg > (uint32_t)MSM_SEG_FAN) {                 // tree over the segment sums: chains of <= 4 additions
                const uint32_t nout = (nseg + MSM_SEG_FAN - 1) / MSM_SEG_FAN;
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_seg_reduce<F>), dim3(nblk(nout, 64), p.nwin), dim3(64), 0, st, cur, nseg, nxt, nout);
                std::swap(cur, nxt); nseg = nout;
            }
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_finish<F>), dim3(1), dim3(64), 0, st, p, cur, nseg, ms.win.as<Jac<F>>(), ms.out.as<Jac<F>>());
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ms.host_out, ms.out.p, sizeof(Jac<F>), hipMemcpyDeviceToHost, st));
        return RIPP_OK;
    }
    template <class F> int32_t msm_dev(const Affine<F>* bases, const Fr* scalars, size_t n, Jac<F>* out_host, const std::function<int32_t()>* bases_arrive = nullptr) {
        int32_t rc = msm_launch<F>(msm_scratch[0], stream, bases, scalars, n, bases_arrive); if (rc) return rc;
        if ((rc = sync())) return rc;
        *out_host = *reinterpret_cast<const Jac<F>*>(msm_scratch[0].host_out);
        return RIPP_OK;
    }

    // ---- pairing product: per-step products of `nprod` products of M pairs each ---------------------------
    // a[p], b[p]: device pointers to M affine pairs for product p.  rows_out: host, [nprod][68] Fp12 (Montgomery).
    // One batch of `nprod` pairing products over the pairs [off, off + m): stage 1 (lines), stage 2 (per-step products + tree) and the copy of
    // the nprod * 68 per-step values into `dst_pinned`, all ENQUEUED on `stream` -- no synchronisation.
    int32_t enqueue_products(const G1A* const* a, const G2A* const* b, int nprod, size_t off, size_t m, Fp12* dst_pinned) {
        const size_t nrows = (size_t)nprod * N_LINES;
        int32_t rc;
        const size_t stride = (m + 63) & ~(size_t)63;
        if ((rc = lines.reserve(nrows * LINE_CHUNKS * stride * sizeof(uint4))) != RIPP_OK) return rc;
        // stage 1: one launch, grid.y = product
        {
            PairSets ps{}; for (int p = 0; p < nprod; ++p) { ps.a[p] = a[p] + off; ps.b[p] = b[p] + off; }
            if ((rc = mark(ev_lines, true)) != RIPP_OK) return rc;
            if (m * nprod <= vm_lines_max && !sw.no_vm)
                hipLaunchKernelGGL(k_vm_miller_lines, dim3(nblk(m, 4 * VM_EPW), nprod), dim3(256), 4 * VM_EPW * VM_LINES_SLOTS * sizeof(VmSlot), stream, ps, (uint32_t)m, lines.as<uint4>(), stride);
            else if (!sw.no_fq && m * nprod >= ml_fq_min) {       // the carry-free twin (fq_miller.hpp): ~25 % fewer instructions on an issue-bound kernel
                // consecutive products over the SAME Q vector share one chain (up to MAX_SHARE P's per lane): the callers order their product lists accordingly
                ChainSets cs{}; int ng = 0;
                for (int p = 0; p < nprod; ++p) {
                    cs.a[p] = ps.a[p];
                    if (!sw.no_share && ng > 0 && ps.b[p] == cs.b[ng - 1] && cs.np[ng - 1] < MAX_SHARE) ++cs.np[ng - 1];
                    else { cs.b[ng] = ps.b[p]; cs.first[ng] = (uint8_t)p; cs.np[ng] = 1; ++ng; }
                }
                hipLaunchKernelGGL(k_miller_lines_q, dim3(nblk(m, 256), ng), dim3(256), 0, stream, cs, (uint32_t)m, lines.as<uint4>(), stride);
                stats.chains_lines += m * (size_t)ng;
            } else
                hipLaunchKernelGGL(k_miller_lines, dim3(nblk(m, 256), nprod), dim3(256), 0, stream, ps, (uint32_t)m, lines.as<uint4>(), stride);
            HIPCHK(hipGetLastError());
            if ((rc = mark(ev_lines, false)) != RIPP_OK) return rc;
            stats.pairs_lines += m * nprod;
        }
        // stage 2 per PIECE of the product list.  A row (product, step) gets T accumulators and the launch has ceil(T / 21) * rows waves, which should
        // fill the chip's 2 waves per SIMD exactly once: rows = 68 * {1, 2, 3, 5, 6} products leave <= 0.4 % of the slots empty, 4 or 7 products 7 %,
        // EIGHT products 20 % (3 * 544 = 1 632 of 2 048 slots: k_line_products_q then takes 25 % longer per pair -- measured on the first shared
        // round-0 launches of build round 4, where it ate the whole gain of the shared G2 chains).  Throughput-sized launches of 8 / 7 / 4 products
        // are therefore split 6 + 2 / 5 + 2 / 2 + 2; the lines of all products still come from ONE stage-1 launch.
        // More than 8 products (a deep look-ahead item: 16, 32 or 64 block products): greedily the piece sizes k for which 68 k rows x an integer number of
        // waves per row fill the 2 048 slots (30 -> 1 wave per row, 15 -> 2, 10 -> 3, 6 -> 5, 5 -> 6, 3 -> 10, 2 -> 15, 1 -> 30): 16 = 15 + 1, 32 = 30 + 2, 64 = 30 + 30 + 3 + 1.
        // (One piece of 16 products left half the chip idle in stage 2: item (2,l) of a 4-rank proof 46 ms where the rate says 34.)
        int pieces[8] = {nprod, 0, 0, 0, 0, 0, 0, 0};
        if (nprod > 8) {
            static const int fill[8] = {30, 15, 10, 6, 5, 3, 2, 1};
            int left = nprod, k = 0;
            while (left > 0 && k < 8) { int f = 0; while (fill[f] > left) ++f; pieces[k++] = fill[f]; left -= fill[f]; }
            if (left > 0) pieces[7] += left;
        }
        else if (m * (size_t)nprod >= ((size_t)1 << 17) && !sw.lp_one_lane) {
            if (nprod == 8) { pieces[0] = 6; pieces[1] = 2; } else if (nprod == 7) { pieces[0] = 5; pieces[1] = 2; } else if (nprod == 4) { pieces[0] = 2; pieces[1] = 2; }
        }
        int p_lo = 0;
        for (int pc = 0; pc < 8 && pieces[pc] > 0; p_lo += pieces[pc], ++pc)
            if ((rc = enqueue_stage2(p_lo, pieces[pc], m, stride, dst_pinned)) != RIPP_OK) return rc;
        return RIPP_OK;
    }
    // stage 2 (per-step products + tree + copy) of the products [p_lo, p_lo + np2) of the line buffer stage 1 has just filled
    int32_t enqueue_stage2(int p_lo, int np2, size_t m, size_t stride, Fp12* dst_pinned) {
        int32_t rc;
        const size_t nrows = (size_t)np2 * N_LINES;
        const uint4* lrows = lines.as<uint4>() + (size_t)p_lo * N_LINES * LINE_CHUNKS * stride;
        // stage 2a: T lanes per row
        // one resident batch: rows * T / 64 waves <= SIMDs * RIPP_OCC, so no partially filled second batch
        // T accumulators per row.  Spill-free form (line_products.hpp): 3 lanes per accumulator, 21 accumulators per wave;
        // RIPP_LP_ONE_LANE=1 selects the one-lane-per-accumulator kernel (A/B builds only).
        // BLS12-381, throughput-sized launches: the Karatsuba form (fq_line_products_k.hpp: six lanes per accumulator, 10 accumulators per wave)
#if !defined(RIPP_BLS12_377)
        const bool lp_kara = !sw.no_fq && !sw.no_lp_kara && !sw.lp_one_lane && m * (size_t)np2 >= lp_fq_min;
#else
        const bool lp_kara = false;
#endif
        const uint32_t per_wave = sw.lp_one_lane ? 64 : lp_kara ? LK_GROUPS_PER_WAVE : LP_GROUPS_PER_WAVE;
        uint32_t T = (uint32_t)std::max<size_t>(per_wave, ((size_t)n_simd * RIPP_OCC_PROD / nrows) * per_wave);
        if (T > m) T = (uint32_t)m;
        if ((rc = partA.reserve(nrows * FP12_CHUNKS * (size_t)T * sizeof(uint4))) != RIPP_OK) return rc;
        if ((rc = partB.reserve(nrows * FP12_CHUNKS * (size_t)((T + 1) / 2) * sizeof(uint4))) != RIPP_OK) return rc;
        if ((rc = mark(ev_prod, true)) != RIPP_OK) return rc;
#if defined(RIPP_AB_KERNELS) && !defined(RIPP_BLS12_377)
        if (sw.lp_one_lane)           // build round 1's one-lane-per-accumulator form: A/B builds only (-DRIPP_AB_KERNELS; it spills 1 248 B per lane)
            hipLaunchKernelGGL(k_line_products1, dim3(nblk(T, 64), (unsigned)nrows), dim3(64), 0, stream, lrows, stride, (uint32_t)m, partA.as<uint4>(), T);
        else
#endif
#if !defined(RIPP_BLS12_377)
        if (lp_kara)
            hipLaunchKernelGGL(k_line_products_k, dim3(nblk(T, LK_GROUPS_PER_WAVE), (unsigned)nrows), dim3(64), 0, stream, lrows, stride, (uint32_t)m, partA.as<uint4>(), T);
        else
#endif
        if (!sw.no_fq && m * (size_t)np2 >= lp_fq_min)          // throughput-sized launches: the carry-free twin (fq_line_products.hpp), both curves
            hipLaunchKernelGGL(k_line_products_q, dim3(nblk(T, LP_GROUPS_PER_WAVE), (unsigned)nrows), dim3(64), 0, stream, lrows, stride, (uint32_t)m, partA.as<uint4>(), T);
        else
            hipLaunchKernelGGL(k_line_products, dim3(nblk(T, LP_GROUPS_PER_WAVE), (unsigned)nrows), dim3(64), 0, stream, lrows, stride, (uint32_t)m, partA.as<uint4>(), T);
        HIPCHK(hipGetLastError());
        if ((rc = mark(ev_prod, false)) != RIPP_OK) return rc;
        stats.pairs_products += m * (size_t)np2;
        // stage 2b: dense tree, radix 4
        uint4* cur = partA.as<uint4>(); uint4* nxt = partB.as<uint4>();
        while (T > 1) {
            // radix 4 while the level still fills the chip, radix 2 (one dependent Fp12 product per level) once it is latency-bound
            const int R = ((size_t)T * nrows > (size_t)n_simd * 64) ? 4 : 2;
            const uint32_t Tout = (T + R - 1) / R;
            if (R == 2 && (size_t)Tout * nrows <= vm_tree_max && !sw.no_vm)
                hipLaunchKernelGGL(k_vm_fp12_tree, dim3(nblk(Tout, 2 * VM_EPW), (unsigned)nrows), dim3(128), 2 * VM_EPW * VM_F12_SLOTS * sizeof(VmSlot), stream, cur, T, nxt, Tout);
            else
            hipLaunchKernelGGL(k_fp12_tree, dim3(nblk(Tout, 64), (unsigned)nrows), dim3(64), 0, stream, cur, T, nxt, Tout, R);
            HIPCHK(hipGetLastError());
            std::swap(cur, nxt); T = Tout;
        }
        // rows are now [nrows][36][1] == nrows contiguous Fp12
        HIPCHK(hipMemcpyAsync(dst_pinned + (size_t)p_lo * N_LINES, cur, nrows * sizeof(Fp12), hipMemcpyDeviceToHost, stream));
        return RIPP_OK;
    }
    int32_t step_products(const G1A* const* a, const G2A* const* b, int nprod, size_t M, Fp12* rows_out) {
        const size_t nrows = (size_t)nprod * N_LINES;
        for (size_t r = 0; r < nrows; ++r) rows_out[r] = Fp12::one();
        if (M == 0) return RIPP_OK;
        int32_t rc;
        if ((rc = ensure_pinned_rows(nrows)) != RIPP_OK) return rc;
        if (nprod > MAX_PRODUCTS) return RIPP_ERR_ARG;
        const size_t batch = std::min(M, std::max<size_t>(1, pairs_cap(std::min(max_pairs_per_batch, M * (size_t)nprod)) / nprod));
        for (size_t off = 0; off < M; off += batch) {
            const size_t m = std::min(batch, M - off);
            if ((rc = enqueue_products(a, b, nprod, off, m, pinned_rows)) != RIPP_OK) return rc;
            if ((rc = sync()) != RIPP_OK) return rc;
            for (size_t r = 0; r < nrows; ++r) rows_out[r] = (off == 0) ? pinned_rows[r] : mul(rows_out[r], pinned_rows[r]);
        }
        return RIPP_OK;
    }
};

int32_t get_engine(Engine** out) {
    if (!g_engine) {
        Engine* e = new Engine();
        int32_t rc = e->init(0);
        if (rc != RIPP_OK) { delete e; return rc; }
        g_engine = e;
    }
    if (hipSetDevice(g_engine->device) != hipSuccess) { set_err("hipSetDevice failed"); return RIPP_ERR_DEVICE; }
    g_engine->refresh_switches();
    g_call_epoch.fetch_add(1, std::memory_order_relaxed); t_engine = g_engine;
    *out = g_engine; return RIPP_OK;
}

bool trace_on() { static const bool on = std::getenv("RIPP_TRACE") != nullptr; return on; }
// frees the scratch of engine e that the current call has not touched; only buffers whose contents no later step of a call takes for granted (each is re-reserved --
// and re-filled -- by the call that uses it; the cached fold tables are disowned so that their next user rebuilds them).  Returns the bytes freed.
size_t evict_idle_scratch(Engine* e) {
    const uint64_t now = g_call_epoch.load(std::memory_order_relaxed);
    (void)hipDeviceSynchronize();
    const size_t before = g_dev_bytes.load();
    bool tables = false;
    for (DevBuf* b : {&e->lines, &e->partA, &e->partB, &e->qtab, &e->scale_tab, &e->jacG1, &e->jacG2, &e->affG1, &e->affG2, &e->tmpA, &e->tmpB, &e->tmpR}) if (b->p && b->epoch != now) b->release();
    for (DevBuf* b : {&e->fold_tab1, &e->fold_mult, &e->fold_tab, &e->fold_jac1, &e->fold_jac2}) if (b->p && b->epoch != now) { b->release(); tables = true; }
    if (tables) { e->tab_owner = nullptr; e->g2tab_hi = nullptr; }
    for (int k = 0; k < 2; ++k) {
        MsmScratch& ms = e->msm_scratch[k];
        for (DevBuf* b : {&ms.digits, &ms.hist, &ms.offs, &ms.cursor, &ms.slotoffs, &ms.spw, &ms.sorted, &ms.slots, &ms.buckets, &ms.seg, &ms.seg2, &ms.win, &ms.ext, &ms.flags}) if (b->p && b->epoch != now) b->release();
        for (DevBuf* b : {&e->kzg_q[k], &e->kzg_bases[k]}) if (b->p && b->epoch != now) b->release();
    }
    if (e->job_cache.full) e->job_cache.release();                        // buffers a finished one-shot proof parked for the next one
    const size_t after = g_dev_bytes.load();
    return before > after ? before - after : 0;
}

ScalarBits scalar_bits(const Fr& s_mont) {
    const Fr c = from_mont(s_mont);
    ScalarBits sb; int top = -1;
    for (int i = 0; i < 8; ++i) { sb.w[i] = c.l[i]; }
    for (int i = 255; i >= 0; --i) if ((c.l[i >> 5] >> (i & 31)) & 1u) { top = i; break; }
    sb.nbits = top + 1; return sb;
}

// non-adjacent form of a little-endian multi-word integer; returns the digit count (digits[0] = least significant)
int naf_recode(const uint32_t* words, int nwords, int8_t* digits, int maxd) {
    uint32_t w[10] = {0};
    for (int i = 0; i < nwords; ++i) w[i] = words[i];
    auto is_zero = [&]() { for (int i = 0; i <= nwords; ++i) if (w[i]) return false; return true; };
    int n = 0;
    while (!is_zero() && n < maxd) {
        int8_t d = 0;
        if (w[0] & 1u) {
            d = (int8_t)(2 - (int)(w[0] & 3u));                // +1 if w = 1 mod 4, -1 if w = 3 mod 4
            if (d > 0) { w[0] &= ~1u; }
            else { for (int i = 0; i <= nwords; ++i) { if (++w[i] != 0) break; } }   // w += 1
        }
        digits[n++] = d;
        for (int i = 0; i < nwords; ++i) w[i] = (w[i] >> 1) | (w[i + 1] << 31);       // w >>= 1
        w[nwords] >>= 1;
    }
    return n;
}
NafDigits naf_digits(const Fr& s_mont) {
    const Fr c = from_mont(s_mont);
    NafDigits nd; std::memset(&nd, 0, sizeof nd);
    nd.len = naf_recode(c.l, 8, nd.d, 258);
    return nd;
}
// s = d0 + d1 u + d2 u^2 + d3 u^3 with u = |x| = 0xd201000000010000, each digit NAF-recoded
GlsDigits gls_digits(const Fr& s_mont) {
    const Fr c = from_mont(s_mont);
    uint64_t v[4] = {(uint64_t)c.l[0] | ((uint64_t)c.l[1] << 32), (uint64_t)c.l[2] | ((uint64_t)c.l[3] << 32),
                     (uint64_t)c.l[4] | ((uint64_t)c.l[5] << 32), (uint64_t)c.l[6] | ((uint64_t)c.l[7] << 32)};
    GlsDigits g; std::memset(&g, 0, sizeof g);
    int maxlen = 0;
    for (int j = 0; j < 4; ++j) {
        unsigned __int128 rem = 0;                             // v <- v / u, digit = v mod u
        for (int i = 3; i >= 0; --i) { const unsigned __int128 cur = (rem << 64) | v[i]; v[i] = (uint64_t)(cur / BLS_X_ABS); rem = cur % BLS_X_ABS; }
        const uint64_t dj = (uint64_t)rem;
        const uint32_t words[2] = {(uint32_t)dj, (uint32_t)(dj >> 32)};
        const int len = naf_recode(words, 2, g.d[j], 66);
        if (len > maxlen) maxlen = len;
    }
    g.len = maxlen;
    return g;
}

// s = s1 + s2 * lambda (lambda = u^2 - 1 ~ sqrt(r)) by binary long division of the canonical scalar; both halves NAF-recoded
void glv_split(const Fr& s_mont, uint32_t rem[9], uint32_t quo[8]) {       // s = rem + quo * lambda, both < 2^128
    const Fr c = from_mont(s_mont);
    const uint32_t lam[8] = RIPP_GLV_LAMBDA;
    for (int i = 0; i < 9; ++i) rem[i] = 0; for (int i = 0; i < 8; ++i) quo[i] = 0;
    for (int bit = 255; bit >= 0; --bit) {                      // rem = rem * 2 + bit;  if rem >= lambda: rem -= lambda, quotient bit = 1
        for (int i = 8; i > 0; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 31);
        rem[0] = (rem[0] << 1) | ((c.l[bit >> 5] >> (bit & 31)) & 1u);
        bool ge = rem[8] != 0;
        if (!ge) { ge = true; for (int i = 7; i >= 0; --i) { if (rem[i] != lam[i]) { ge = rem[i] > lam[i]; break; } } }
        if (ge) { uint32_t borrow = 0; for (int i = 0; i < 8; ++i) rem[i] = subb32(rem[i], lam[i], borrow); rem[8] -= borrow; quo[bit >> 5] |= 1u << (bit & 31); }
    }
}
GlvDigits glv_digits(const Fr& s_mont) {
    uint32_t rem[9], quo[8]; glv_split(s_mont, rem, quo);
    GlvDigits g; std::memset(&g, 0, sizeof g);
    const int l1 = naf_recode(rem, 5, g.d1, 131), l2 = naf_recode(quo, 5, g.d2, 131);
    g.len = l1 > l2 ? l1 : l2;
    return g;
}

// digit strings of the folds with a precomputed second base (kernels.hpp, "round-0 folds")
// width-w wNAF (w = RIPP_FOLD_W) of a value < 2^64: odd digits of magnitude < 2^(w-1), at most one nonzero in any w consecutive positions
int wnaf4_recode(uint64_t v, int8_t* digits, int maxd, int W = RIPP_FOLD_W) {
    unsigned __int128 k = v; int len = 0;
    while (k != 0 && len < maxd) {
        int d = 0;
        if ((uint64_t)k & 1u) { d = (int)((uint64_t)k & ((1u << W) - 1)); if (d >= (1 << (W - 1))) d -= 1 << W; if (d >= 0) k -= (unsigned)d; else k += (unsigned)(-d); }
        digits[len++] = (int8_t)d;
        k >>= 1;
    }
    return len;
}
Wnaf4 split32_wnaf(const Fr& s_mont, int W = RIPP_FOLD_W) {    // 128-bit challenge -> width-W wNAF strings of its four 32-bit words
    const Fr c = from_mont(s_mont);
    Wnaf4 g; std::memset(&g, 0, sizeof g);
    for (int t = 0; t < 4; ++t) g.len = std::max(g.len, wnaf4_recode(c.l[t], g.d[t], 35, W));
    return g;
}
int tab_width(int M) { int w = 2; while ((1 << (w - 2)) < M) ++w; return w; }      // M = 2^(W - 2) odd multiples per base <-> wNAF width W
void glv_split(const Fr& s_mont, uint32_t rem[9], uint32_t quo[8]);
// the sixteen strings of the fused G1 fold (fq_curve.hpp k_fold_g1_fused_q): x0 | k1 | k2 | x1 with x0 x1 = k1 + k2 lambda
WnafG1x4 fused_digits_g1(const Fr& x0, const Fr& x1, int W) {
    WnafG1x4 g; std::memset(&g, 0, sizeof g);
    const Fr c0 = from_mont(x0), c1 = from_mont(x1);
    uint32_t rem[9], quo[8]; glv_split(mul(x0, x1), rem, quo);
    const uint32_t* src[4] = {c0.l, rem, quo, c1.l};
    for (int u = 0; u < 4; ++u) for (int b = 0; b < 4; ++b) g.len = std::max(g.len, wnaf4_recode(src[u][b], g.d[4 * u + b], 35, W));
    return g;
}
GlvDigits split64_digits(const Fr& s_mont) {                     // 128-bit challenge -> its two 64-bit halves
    const Fr c = from_mont(s_mont);
    GlvDigits g; std::memset(&g, 0, sizeof g);
    const int l1 = naf_recode(&c.l[0], 2, g.d1, 131), l2 = naf_recode(&c.l[2], 2, g.d2, 131);
    g.len = l1 > l2 ? l1 : l2;
    return g;
}
bool fits_128(const Fr& s_mont) { const Fr c = from_mont(s_mont); return (c.l[4] | c.l[5] | c.l[6] | c.l[7]) == 0; }
Gls8Digits gls8_digits(const Fr& s_mont) {                        // base-u digits (u = |x|), each split at bit 32
    const Fr c = from_mont(s_mont);
    uint64_t v[4] = {(uint64_t)c.l[0] | ((uint64_t)c.l[1] << 32), (uint64_t)c.l[2] | ((uint64_t)c.l[3] << 32),
                     (uint64_t)c.l[4] | ((uint64_t)c.l[5] << 32), (uint64_t)c.l[6] | ((uint64_t)c.l[7] << 32)};
    Gls8Digits g; std::memset(&g, 0, sizeof g);
    int maxlen = 0;
    for (int j = 0; j < 4; ++j) {
        unsigned __int128 rem = 0;
        for (int i = 3; i >= 0; --i) { const unsigned __int128 cur = (rem << 64) | v[i]; v[i] = (uint64_t)(cur / BLS_X_ABS); rem = cur % BLS_X_ABS; }
        const uint64_t dj = (uint64_t)rem;
        const uint32_t lo = (uint32_t)dj, hi = (uint32_t)(dj >> 32);
        const int l1 = naf_recode(&lo, 1, g.d[j], 35), l2 = naf_recode(&hi, 1, g.d[4 + j], 35);
        maxlen = std::max(maxlen, std::max(l1, l2));
    }
    g.len = maxlen;
    return g;
}

Wnaf16 gls16_wnaf(const Fr& s_mont, int W = RIPP_FOLD_W) {     // base-u digits, each cut into four 16-bit pieces, width-W wNAF strings
    const Fr c = from_mont(s_mont);
    uint64_t v[4] = {(uint64_t)c.l[0] | ((uint64_t)c.l[1] << 32), (uint64_t)c.l[2] | ((uint64_t)c.l[3] << 32),
                     (uint64_t)c.l[4] | ((uint64_t)c.l[5] << 32), (uint64_t)c.l[6] | ((uint64_t)c.l[7] << 32)};
    Wnaf16 g; std::memset(&g, 0, sizeof g);
    for (int j = 0; j < 4; ++j) {
        unsigned __int128 rem = 0;
        for (int i = 3; i >= 0; --i) { const unsigned __int128 cur = (rem << 64) | v[i]; v[i] = (uint64_t)(cur / BLS_X_ABS); rem = cur % BLS_X_ABS; }
        const uint64_t dj = (uint64_t)rem;
        for (int b = 0; b < 4; ++b) g.len = std::max(g.len, wnaf4_recode((dj >> (16 * b)) & 0xffffu, g.d[4 * b + j], 19, W));
    }
    return g;
}
// the three digit sets of the fused G2 fold (fq_curve2.hpp k_fold_g2_fused_q): x0 x1 (full width) | x0 | x1
Wnaf16x3 fused_digits_g2(const Fr& x0, const Fr& x1, int W) {
    Wnaf16x3 g; g.s[0] = gls16_wnaf(mul(x0, x1), W); g.s[1] = gls16_wnaf(x0, W); g.s[2] = gls16_wnaf(x1, W);
    g.len = std::max(g.s[0].len, std::max(g.s[1].len, g.s[2].len));
    return g;
}

GlsDigits gls_wnaf(const Fr& s_mont, int W) {                     // base-u digits as width-W wNAF strings (one base, in-round tables)
    const Fr c = from_mont(s_mont);
    uint64_t v[4] = {(uint64_t)c.l[0] | ((uint64_t)c.l[1] << 32), (uint64_t)c.l[2] | ((uint64_t)c.l[3] << 32),
                     (uint64_t)c.l[4] | ((uint64_t)c.l[5] << 32), (uint64_t)c.l[6] | ((uint64_t)c.l[7] << 32)};
    GlsDigits g; std::memset(&g, 0, sizeof g);
    for (int j = 0; j < 4; ++j) {
        unsigned __int128 rem = 0;
        for (int i = 3; i >= 0; --i) { const unsigned __int128 cur = (rem << 64) | v[i]; v[i] = (uint64_t)(cur / BLS_X_ABS); rem = cur % BLS_X_ABS; }
        g.len = std::max(g.len, wnaf4_recode((uint64_t)rem, g.d[j], 66, W));
    }
    return g;
}

// digit strings for vm_fold2.hpp: the same splits as split64_digits / gls8_digits in the SplitDigits layout
SplitDigits split_digits_g1(const Fr& s_mont) {
    const GlvDigits g = split64_digits(s_mont);
    SplitDigits d; std::memset(&d, 0, sizeof d);
    for (int i = 0; i < g.len && i < 68; ++i) { d.d[0][i] = g.d1[i]; d.d[1][i] = g.d2[i]; }
    d.len = g.len; return d;
}
SplitDigits split_digits_g2(const Fr& s_mont) {
    const Gls8Digits g = gls8_digits(s_mont);
    SplitDigits d; std::memset(&d, 0, sizeof d);
    for (int t = 0; t < 8; ++t) for (int i = 0; i < g.len; ++i) d.d[t][i] = g.d[t][i];
    d.len = g.len; return d;
}

// full-width G1 scalar for the second-base VM fold: GLV halves k1, k2 (< 2^128), each split at bit 64 ->
// d[0] = k1_lo (P), d[1] = k2_lo (phi P), d[2] = k1_hi (2^64 P), d[3] = k2_hi (phi 2^64 P)
SplitDigits split_digits_g1_glv(const Fr& s_mont) {
    uint32_t rem[9], quo[8]; glv_split(s_mont, rem, quo);
    SplitDigits d; std::memset(&d, 0, sizeof d);
    const int l0 = naf_recode(&rem[0], 2, d.d[0], 67), l1 = naf_recode(&quo[0], 2, d.d[1], 67), l2 = naf_recode(&rem[2], 2, d.d[2], 67), l3 = naf_recode(&quo[2], 2, d.d[3], 67);
    d.len = std::max(std::max(l0, l1), std::max(l2, l3)); return d;
}

template <class T> int32_t upload(Engine* e, DevBuf& buf, const void* host, size_t n, T** dev) {
    int32_t rc = buf.reserve(std::max<size_t>(n, 1) * sizeof(T)); if (rc != RIPP_OK) return rc;
    if (n) HIPCHK(hipMemcpyAsync(buf.p, host, n * sizeof(T), hipMemcpyHostToDevice, e->stream));
    *dev = buf.as<T>(); return RIPP_OK;
}


// Blake2s of (a, b, r, value).serialize_uncompressed (sipp/src/lib.rs:56-59).  Serialisation (Montgomery -> canonical
// big-endian) is spread over worker threads in blocks; the hash itself is inherently sequential.
double g_digest_hash_ms = 0, g_digest_wait_ms = 0;
void statement_digest(const G1A* a, const G2A* b, const Fr* r, size_t n, const Fp12& value, uint8_t digest[32], std::atomic<uint64_t>* progress = nullptr) {
    fs::Blake2s h;
    g_digest_hash_ms = g_digest_wait_ms = 0;
    const uint64_t len = (uint64_t)n;
    // The hash is sequential; the Montgomery -> canonical big-endian serialisation feeding it is not.  Segments of BLK items are
    // serialised by the persistent host workers into a ring of buffers, RING - 1 segments ahead of the hash.
    constexpr size_t BLK = 1 << 13; constexpr int RING = 6;
    const size_t item[3] = {96, 192, 32};
    struct Seg { int kind; size_t s, e; };
    std::vector<Seg> segs;
    for (int kind = 0; kind < 3; ++kind) for (size_t s0 = 0; s0 < n; s0 += BLK) segs.push_back({kind, s0, std::min(n, s0 + BLK)});
    // The ring is taken from a process-wide pool and NEVER freed: allocating and releasing 9 MB of host memory per call makes glibc map /
    // trim it per call, and an unmap next to (same 2 MB huge page as) a caller array the HIP runtime has pinned for an upload invalidates
    // that registration -- the driver then evicts and restores the process's GPU queues, which stalled the next device operation of the
    // n = 2^13..2^14 verifier by 15-40 ms in steps of 10 ms in most processes (tools/kdev/verify_lat2.py).
    struct Ring { std::vector<uint8_t> b[RING]; };
    static std::mutex pool_mu; static std::vector<std::unique_ptr<Ring>> pool;
    std::unique_ptr<Ring> ring;
    { std::lock_guard<std::mutex> lk(pool_mu); if (!pool.empty()) { ring = std::move(pool.back()); pool.pop_back(); } }
    if (!ring) { ring.reset(new Ring()); for (auto& v : ring->b) v.resize(BLK * 192); }
    struct Return { std::unique_ptr<Ring>& r; std::mutex& m; std::vector<std::unique_ptr<Ring>>& p; ~Return() { std::lock_guard<std::mutex> lk(m); p.emplace_back(std::move(r)); } } give_back{ring, pool_mu, pool};
    std::vector<uint8_t>* const buf = ring->b;
    std::vector<std::future<void>> fut(segs.size());
    auto launch = [&](size_t j) {
        const Seg sg = segs[j]; uint8_t* out = buf[j % RING].data();
        fut[j] = host_pool().submit([a, b, r, sg, out]() {
            for (size_t i = sg.s; i < sg.e; ++i) {
                if (sg.kind == 0) fs::ser_g1(a[i], out + (i - sg.s) * 96);
                else if (sg.kind == 1) fs::ser_g2(b[i], out + (i - sg.s) * 192);
                else fs::ser_fr(r[i], out + (i - sg.s) * 32);
            } });
    };
    for (size_t j = 0; j < segs.size() && j < (size_t)RING - 1; ++j) launch(j);
    int cur_kind = -1;
    for (size_t j = 0; j < segs.size(); ++j) {
        if (segs[j].kind != cur_kind) {      // a new vector starts: its u64 length prefix (ark-serialize Vec), also for empty vectors below
            for (int k = cur_kind + 1; k <= segs[j].kind; ++k) h.update(reinterpret_cast<const uint8_t*>(&len), 8);
            cur_kind = segs[j].kind;
        }
        const double t0 = now_ms();
        fut[j].get();
        const double t1 = now_ms();
        h.update(buf[j % RING].data(), (segs[j].e - segs[j].s) * item[segs[j].kind]);
        if (progress) progress->fetch_add((segs[j].e - segs[j].s) * item[segs[j].kind], std::memory_order_relaxed);
        g_digest_wait_ms += t1 - t0; g_digest_hash_ms += now_ms() - t1;
        if (j + RING - 1 < segs.size()) launch(j + RING - 1);          // its buffer was released by the segment just hashed... one slot later
    }
    for (int k = cur_kind + 1; k < 3; ++k) h.update(reinterpret_cast<const uint8_t*>(&len), 8);       // n == 0: three empty vectors
    uint8_t gt[576]; fs::ser_gt(value, gt); h.update(gt, 576);
    h.finish(digest);
}

}  // namespace

// =============================================================================================== SIPP job
struct ripp_sipp_job {
    size_t n_local = 0, len = 0;          // statement shard size; current vector length of this shard
    int rank = 0, world = 1, world0 = 1;  // world0: sharding of the resident statement; world drops to 1 after the tail import
    DevBuf a0, b0, r0;                    // resident statement shard
    DevBuf a, b, a_next, b_next, jac1, jac2;   // working vectors
    size_t tab2_stride = 0; bool tab_ready = false;   // round-0 tables (odd multiples of four bases) of this job are in the engine's fold_* buffers
    // tab_fused: the tables cover THREE quarters (A1 | A2 | A3 and B0 | B1 | B2: tab_cnt = 3 len / 4 elements per row, tab_M odd multiples per base) so that
    // rounds 0 and 1 can be folded in one pass once both challenges are known (job_fold_fused); otherwise a_r / b_l (b_r), tab_cnt = len / 2
    size_t tab_cnt = 0; int tab_M = FOLD_TAB_M; bool tab_fused = false;
    // x-SCALED G2 vector (single-GPU prover, large rounds): the device holds bt = bs * b instead of b, folded as bt' = x bt_l + bt_r with the
    // 128-bit challenge x (two GLS digit strings instead of four: ~36 % less G2 fold work); the round's two GT values are then z^bs and
    // the host takes them to the power 1/bs.  The first round below the table-fold size returns to the plain vector in its own fold.
    Fr bs = Fr::one(); bool bs_on = false; bool tab_on_lo = false; bool xs_enabled = false;      // xs_enabled: set by ripp_sipp_job_prove only
    DevBuf a_pow, b_pow; bool pre_ready = false;   // 2^64 * a_r and 2^32 * b_r of round 0, prepared while the statement hash finishes
    // values of rounds 1..k pre-evaluated in the hash window through bilinearity (see job_lookahead): one item per (round, side)
    struct LookItem {
        int R = 0, side = 0, level = 0;            // round whose value this gives; 0 = z_l, 1 = z_r; challenges x_0 .. x_(level-1) already applied
        size_t npairs = 0, q = 0;                   // pairs [0, npairs) of every block (of q pairs) are covered; a PARTIAL item (npairs < q) leaves pairs [npairs, q) of that round's product to the device
        std::vector<Fp12> Z;                        // 3^(R - level) GT values, index = sum_t (d_t + 1) 3^(R-1-t) over the challenges still to come
        std::vector<std::future<Fp12>> fe, pend;    // final exponentiations still running; GT powers of the level being applied (3 per output)
    };
    std::vector<LookItem> look; PinBuf look_rows[2];
    double look_deadline = 0;      // ranks that do NOT hash: the time (now_ms) at which rank 0 expects the digest -- they size their look-ahead against it with their OWN measured pairing rate
    std::atomic<uint64_t> hash_done{0}; uint64_t hash_total = 0; double hash_t0 = 0;      // progress of the statement hash (bytes), for the adaptive look-ahead
    bool no_window = false;                         // sharded proofs: rank 0 was handed the digest, nobody hashes, nothing to hide work behind
    size_t hash_n = 0;                              // length of the statement ha_ext / hb_ext / hr_ext point to (the FULL statement on rank 0 of a sharded proof)
    DevBuf a_pow_h, b_pow_h, parts1, parts2; bool pre_vm_ready = false, pre_vm_side = false;   // the same for the small rounds, on the field VM (vm_fold2.hpp); side: b_pow_h comes from stream3 (ev_join3)
    // pipelined tail (job_tail_enqueue): per-step values of the eight quarter products that give round R's (z_l, z_r) once x_(R-1) is known,
    // evaluated from round R-1's UNFOLDED vectors; slot R & 1, tp_round[slot] = R while they are enqueued / waiting to be used
    PinBuf tp_rows[2]; size_t tp_round[2] = {~(size_t)0, ~(size_t)0};
    std::vector<G1A> ha; std::vector<G2A> hb; std::vector<Fr> hr;   // host copy of the statement (rank 0 hashes it)
    const G1A* ha_ext = nullptr; const G2A* hb_ext = nullptr; const Fr* hr_ext = nullptr;   // one-shot proofs hash the CALLER's buffers in place
    bool hash_prestarted = false;
    fs::FiatShamirRng rng; bool seeded = false;
    std::thread hash_thread; uint8_t digest[32]; std::atomic<bool> digest_ready{false};
    double t_begin = 0;
};

namespace {

int32_t job_begin(Engine* e, ripp_sipp_job* j) {
    const size_t n = j->n_local;
    int32_t rc;
    if ((rc = j->a.reserve(n * sizeof(G1A))) != RIPP_OK) return rc;
    if ((rc = j->b.reserve(n * sizeof(G2A))) != RIPP_OK) return rc;
    if ((rc = j->a_next.reserve(std::max<size_t>(n / 2, 1) * sizeof(G1A))) != RIPP_OK) return rc;
    if ((rc = j->b_next.reserve(std::max<size_t>(n / 2, 1) * sizeof(G2A))) != RIPP_OK) return rc;
    if ((rc = j->jac1.reserve(n * sizeof(G1J))) != RIPP_OK) return rc;
    if ((rc = j->jac2.reserve(std::max<size_t>(n / 2, 1) * sizeof(G2J))) != RIPP_OK) return rc;
    e->stats = ripp_stats{};
    j->t_begin = now_ms();
    // a_i <- r_i * a_i, normalised (sipp/src/lib.rs:61-66); b copied (:67)
    hipEvent_t t0 = e->ev_t0, t1 = e->ev_t1;
    HIPCHK(hipEventRecord(t0, e->stream));
    if ((rc = e->scale_g1_dev(j->a0.as<G1A>(), 1u, j->r0.as<Fr>(), n, j->jac1.as<G1J>())) != RIPP_OK) return rc;
    if ((rc = e->normalize_dev<Fp>(j->jac1.as<G1J>(), n, j->a.as<G1A>())) != RIPP_OK) return rc;
    HIPCHK(hipMemcpyAsync(j->b.p, j->b0.p, n * sizeof(G2A), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipEventRecord(t1, e->stream));
    if ((rc = e->sync()) != RIPP_OK) return rc;
    float ms = 0; (void)hipEventElapsedTime(&ms, t0, t1); e->stats.scale_ms += ms;
    j->len = n; j->seeded = false; j->world = j->world0;
    j->bs = Fr::one(); j->bs_on = false;
    j->pre_vm_ready = false; j->pre_vm_side = false; j->pre_ready = false; j->look.clear(); j->tp_round[0] = j->tp_round[1] = ~(size_t)0;      // nothing prepared for these vectors yet
    return RIPP_OK;
}

// this shard's per-step products for z_l = prod e(a_r, b_l), z_r = prod e(a_l, b_r)   (sipp/src/lib.rs:70-78)
int32_t job_round_partials(Engine* e, ripp_sipp_job* j, Fp12* rows /* [2][68] */) {
    const size_t half = j->len / 2;
    const G1A* a = j->a.as<G1A>(); const G2A* b = j->b.as<G2A>();
    const G1A* as[2] = {a + half, a};       // z_l pairs a_r with b_l; z_r pairs a_l with b_r
    const G2A* bs[2] = {b, b + half};
    const double t0 = now_ms();
    int32_t rc = e->step_products(as, bs, 2, half, rows);
    e->stats.miller_products_ms += now_ms() - t0;
    return rc;
}

bool fold_g2_table_pays(const Engine* e, size_t half) { return half >= e->fold_tab_min && half > e->gls_split_max && !e->sw.no_fold_tables; }
// ... and the device can hold them: (M + 4 M) affine rows + (M - 1) Jacobian rows per element, M = 4 (fold_g2_table_build).  NOT part of fold_g2_table_pays:
// that predicate decides whether the vector is x-scaled and must give the same answer every time it is asked within a round; a round whose tables do not
// fit runs the SAME fold (same halves, same scalar) through the 4-lane split form instead (job_fold), whatever the x-scaling.
bool fold_g2_table_fits(const Engine* e, size_t half) {
    const size_t need = half * (5 * 4 * sizeof(G2A) + 3 * sizeof(G2J)) + 65536, held = e->fold_mult.cap + e->fold_tab.cap + e->fold_jac2.cap;
    return need <= held || e->mem_fits(need, held);
}
// rounds whose G2 fold runs on the x-scaled vector (see ripp_sipp_job::bs): the table folds of a proof driven by sipp_prove_core (every rank of a
// sharded proof scales ITS shard: z^bs -> z is a homomorphism, so the ranks' corrected partial values multiply to the same group element)
bool xscale_round(const Engine* e, const ripp_sipp_job* j, size_t half) {
    if (!j->xs_enabled || e->sw.no_endo || e->sw.no_xscale) return false;
    if (fold_g2_table_pays(e, half)) return true;
    return j->bs_on && half <= e->gls_split_max && 2 * half > e->vm_joint_max && half > e->vm_fold_max;      // stays scaled down to the largest joint-VM round: the return is cheapest on the smallest VM round that walks one group per element
}
// Round 0 only, single-GPU proofs: enqueue hi2 = 2^64 a_r and 2^32 b_r (normalised) behind the round's pairing products.  The GPU would
// otherwise idle until the statement hash delivers the first challenge; the fold then needs half the doublings (k_fold_g1_two /
// k_fold_g2_gls8).  Not worth it for small rounds (latency-bound) -- and skipped when the hash is already done.
// Small rounds (VM regime), every round: the same second bases on the field VM, projective (no normalisation), enqueued behind the
// round's pairing products so that they run during the host's final exponentiations.
// side: (pipelined tail, called right after an un-waited fold) the G2 doublings go to stream3 behind that fold, so that the quarter products
// enqueued next on the main stream run beside them instead of after them; the next fold waits for ev_join3.
int32_t job_precompute_vm(Engine* e, ripp_sipp_job* j, bool side = false) {
    const size_t half = j->len / 2;
    if (half == 0 || half > e->vm_fold_max || e->sw.no_vm || e->sw.no_precompute) return RIPP_OK;
    int32_t rc;
    if ((rc = j->a_pow_h.reserve(half * sizeof(G1J))) || (rc = j->b_pow_h.reserve(half * sizeof(G2J))) || (rc = j->parts1.reserve(2 * half * sizeof(G1J))) || (rc = j->parts2.reserve(8 * half * sizeof(G2J)))) return rc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_vm_pow2<Fp>), dim3(nblk(half, 4 * VM_EPW)), dim3(256), 4 * VM_EPW * VmCurve<Fp>::SLOTS * sizeof(VmSlot), e->stream2, j->a.as<G1A>() + half, (uint32_t)half, 64, j->a_pow_h.as<G1J>());
    hipStream_t s2 = e->stream;
    if (side) { HIPCHK(hipStreamWaitEvent(e->stream3, e->ev_t1, 0)); s2 = e->stream3; }      // ev_t1: recorded by job_fold at the end of the fold
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_vm_pow2<Fp2>), dim3(nblk(half, 4 * VM_EPW)), dim3(256), 4 * VM_EPW * VmCurve<Fp2>::SLOTS * sizeof(VmSlot), s2, j->b.as<G2A>() + half, (uint32_t)half, 32, j->b_pow_h.as<G2J>());
    HIPCHK(hipGetLastError());
    if (side) HIPCHK(hipEventRecord(e->ev_join3, e->stream3));
    j->pre_vm_side = side;
    j->pre_vm_ready = true;
    return RIPP_OK;
}

// fuse: the look-ahead is expected to deliver BOTH values of round 1 in the hash window, so x0 and x1 will be known together -- build the tables over
// three quarters (a[q .. 4q) = A1 | A2 | A3, b[0 .. 3q) = B0 | B1 | B2) with FOUR odd multiples per base (width-4 strings): about the table work of
// the half-vector tables with eight, and rounds 0 and 1 then fold in one pass (job_fold_fused).  If x1 is late after all, round 0 folds alone over
// the same tables (element offset q on G1, none on G2).
int32_t job_precompute_round0(Engine* e, ripp_sipp_job* j, bool fuse = false) {
    const size_t half = j->len / 2, q = j->len / 4;
    if (half < ((size_t)1 << 16) || j->digest_ready.load() || j->no_window || e->sw.no_precompute) return RIPP_OK;
    int32_t rc;
    bool tables = !e->sw.no_fold_tables;
    j->tab_ready = false; j->tab_fused = false;
    // What the device can hold decides the form (Engine::mem_fits; every form folds to the same group elements):
    //   three-quarter tables, 4 multiples (13.6 KB per element of the vector)  >  half-vector tables, FOLD_TAB_M multiples (18.4 KB; the plan without
    //   a fused fold)  >  half-vector tables, 4 multiples (9.1 KB)  >  the two pre-doubled bases (0.43 KB)  >  nothing (the folds run their full chains)
    auto tab_bytes = [](size_t M, size_t cnt) { return cnt * (4 * M * (sizeof(G1A) + sizeof(G2A)) + 16 * M * sizeof(G2A) + (M - 1) * (sizeof(G1J) + sizeof(G2J)) + 4) + 65536; };
    const size_t tab_held = e->fold_tab1.cap + e->fold_mult.cap + e->fold_tab.cap + e->fold_jac1.cap + e->fold_jac2.cap + e->fix_flags.cap;
    size_t M_half = FOLD_TAB_M;
    if (tables) {
        const bool want_fuse = fuse && xscale_round(e, j, half) && !e->sw.no_fq && !e->sw.no_fuse && q >= e->fq_min;
        if (want_fuse && !e->mem_fits(tab_bytes(4, 3 * q), tab_held)) { fuse = false; e->mem_tier = std::max(e->mem_tier & 7, 1) | (e->mem_tier & 8); }
        if (!(want_fuse && fuse)) {
            if (!e->mem_fits(tab_bytes(M_half, half), tab_held)) { M_half = 4; e->mem_tier = std::max(e->mem_tier & 7, 1) | (e->mem_tier & 8); }
            if (!e->mem_fits(tab_bytes(M_half, half), tab_held)) { tables = false; e->mem_tier = std::max(e->mem_tier & 7, 2) | (e->mem_tier & 8); }
        }
        // no room for round-0 tables: the rest of THIS call runs the table-free form as a whole (the RIPP_NO_FOLD_TABLES / no_fold_tables path the 2^17 switch
        // test pins) -- the x-scaled G2 vector and the in-round tables of rounds 1-4 go with them; the switches are re-read at the next C-ABI call
        if (!tables) e->sw.no_fold_tables = true;
    }
    if (!tables && !e->mem_fits(half * (sizeof(G1A) + sizeof(G2A) + sizeof(G1J) + sizeof(G2J)), j->a_pow.cap + j->b_pow.cap + j->jac1.cap + j->jac2.cap)) {
        e->mem_tier = 3 | (e->mem_tier & 8);
        if (trace_on()) fprintf(stderr, "[ripp] round-0 precomputation skipped: the device is short of memory\n");
        return RIPP_OK;
    }
    if (trace_on() && (e->mem_tier & 7)) fprintf(stderr, "[ripp] round-0 tables: memory tier %d (%s)\n", e->mem_tier & 7, tables ? "half-vector tables" : "pre-doubled bases only");
    if ((rc = j->a_pow.reserve(half * sizeof(G1A))) || (rc = j->b_pow.reserve(half * sizeof(G2A))) || (rc = j->jac1.reserve(half * sizeof(G1J))) || (rc = j->jac2.reserve(half * sizeof(G2J)))) return rc;
    if (!tables) {                                              // two-base form: 2^64 a_r, 2^32 b_r
        if (!e->sw.no_fq) hipLaunchKernelGGL(k_pow2_mul_g1_q, dim3(nblk(half, 256)), dim3(256), 0, e->stream, j->a.as<G1A>() + half, (uint32_t)half, 64, j->jac1.as<G1J>());
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pow2_mul<Fp>), dim3(nblk(half, 256)), dim3(256), 0, e->stream, j->a.as<G1A>() + half, (uint32_t)half, 64, j->jac1.as<G1J>());
        HIPCHK(hipGetLastError());
        if ((rc = e->normalize_dev<Fp>(j->jac1.as<G1J>(), half, j->a_pow.as<G1A>()))) return rc;
        if (!e->sw.no_fq) hipLaunchKernelGGL(k_pow2_mul_g2_q, dim3(nblk(half, 64)), dim3(64), 0, e->stream, j->b.as<G2A>() + half, (uint32_t)half, 32, j->jac2.as<G2J>());
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pow2_mul<Fp2>), dim3(nblk(half, 256)), dim3(256), 0, e->stream, j->b.as<G2A>() + half, (uint32_t)half, 32, j->jac2.as<G2J>());
        HIPCHK(hipGetLastError());
        if ((rc = e->normalize_dev<Fp2>(j->jac2.as<G2J>(), half, j->b_pow.as<G2A>()))) return rc;
        j->pre_ready = true;
        return RIPP_OK;
    }
    // table form (kernels.hpp): bases 2^(32 b) P and 2^(16 b) Q, b < 4, and the odd multiples 1, 3, .., 2 M - 1 of each.
    // tab1 / mult2 hold [M b + m][cnt]; row M b is base b itself, written by the doubling chain's normalisation.
    j->tab_on_lo = xscale_round(e, j, half);                    // scaled fold: x multiplies the LOW half
    fuse = fuse && j->tab_on_lo && !e->sw.no_fq && !e->sw.no_fuse && q >= e->fq_min;
    const size_t M = fuse ? 4 : M_half;
    const size_t cnt = fuse ? 3 * q : half;
    const size_t qstride = (cnt + 63) & ~(size_t)63;
    const size_t njt = (M - 1) * cnt;
    if ((rc = e->fold_tab1.reserve(4 * M * cnt * sizeof(G1A))) || (rc = e->fold_mult.reserve(4 * M * cnt * sizeof(G2A))) || (rc = e->fold_tab.reserve(16 * M * G2A_CHUNKS * qstride * sizeof(uint4))) ||
        (rc = e->fold_jac1.reserve(njt * sizeof(G1J))) || (rc = e->fold_jac2.reserve(njt * sizeof(G2J))) || (rc = e->fix_flags.reserve(4 * cnt + 16))) return rc;      // all sized BEFORE the first launch: reserve() may reallocate
    e->tab_owner = j;
    G1A* t1 = e->fold_tab1.as<G1A>(); G2A* t2 = e->fold_mult.as<G2A>();
    G1J* sj1 = e->fold_jac1.as<G1J>(); G2J* sj2 = e->fold_jac2.as<G2J>();
    HIPCHK(hipMemcpyAsync(t1, j->a.as<G1A>() + (fuse ? q : half), cnt * sizeof(G1A), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(t2, j->b.as<G2A>() + (j->tab_on_lo ? 0 : half), cnt * sizeof(G2A), hipMemcpyDeviceToDevice, e->stream));
    for (int b = 0; b < 4; ++b) {
        G1A* base1 = t1 + M * b * cnt; G2A* base2 = t2 + M * b * cnt;
        if (b > 0) {
            if (!e->sw.no_fq) hipLaunchKernelGGL(k_pow2_mul_g1_q, dim3(nblk(cnt, 256)), dim3(256), 0, e->stream, base1 - M * cnt, (uint32_t)cnt, 32, sj1);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pow2_mul<Fp>), dim3(nblk(cnt, 256)), dim3(256), 0, e->stream, base1 - M * cnt, (uint32_t)cnt, 32, sj1);
            HIPCHK(hipGetLastError());
            if ((rc = e->normalize_dev<Fp>(sj1, cnt, base1))) return rc;
            if (!e->sw.no_fq) hipLaunchKernelGGL(k_pow2_mul_g2_q, dim3(nblk(cnt, 64)), dim3(64), 0, e->stream, base2 - M * cnt, (uint32_t)cnt, 16, sj2);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pow2_mul<Fp2>), dim3(nblk(cnt, 256)), dim3(256), 0, e->stream, base2 - M * cnt, (uint32_t)cnt, 16, sj2);
            HIPCHK(hipGetLastError());
            if ((rc = e->normalize_dev<Fp2>(sj2, cnt, base2))) return rc;
        }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_odd_multiples<Fp>), dim3(nblk(cnt, 64)), dim3(64), 0, e->stream, base1, (uint32_t)cnt, (int)M, sj1);
        HIPCHK(hipGetLastError());
        if ((rc = e->normalize_dev<Fp>(sj1, njt, base1 + cnt))) return rc;
        if (!e->sw.no_fq) {
            hipLaunchKernelGGL(k_odd_multiples_q, dim3(nblk(cnt, 64)), dim3(64), 0, e->stream, base2, (uint32_t)cnt, (int)M, sj2, e->fix_flags.as<uint8_t>());
#if !defined(RIPP_INLINE_FALLBACK)
            hipLaunchKernelGGL(k_odd_multiples_fix, dim3(FIX_GRID), dim3(64), 0, e->stream, base2, (uint32_t)cnt, (int)M, sj2, e->fix_flags.as<uint8_t>());
#endif
        }
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_odd_multiples<Fp2>), dim3(nblk(cnt, 64)), dim3(64), 0, e->stream, base2, (uint32_t)cnt, (int)M, sj2);
        HIPCHK(hipGetLastError());
        if ((rc = e->normalize_dev<Fp2>(sj2, njt, base2 + cnt))) return rc;
    }
    hipLaunchKernelGGL(k_g2_tab_images, dim3(nblk(cnt, 64), 4 * M), dim3(64), 0, e->stream, e->fold_mult.as<G2A>(), (uint32_t)cnt, (int)M, e->fold_tab.as<uint4>(), qstride);
    HIPCHK(hipGetLastError());
    j->tab2_stride = qstride; j->tab_cnt = cnt; j->tab_M = (int)M; j->tab_fused = fuse; j->tab_ready = true;
    return RIPP_OK;
}

// G2 fold of a throughput-bound round over in-round tables: odd multiples {1,3,5,7} of every hi element (batch-normalised: the inversion
// is shared by 16 points) and their psi images, then width-4 wNAF strings -- 65 doublings + ~52 additions instead of 65 + ~87 for
// ~350 Fp products of table work per element.  Leaves the Jacobian result in jac (first `half` entries).
// The table half of it depends on the vector only, not on the challenge: fold_g2_table_build may be enqueued before the challenge exists (job_prebuild_g2_tables:
// the device is idle while the host computes the round's final exponentiations), fold_g2_table then finds e->g2tab_hi / g2tab_half set and goes straight to the fold.
int32_t fold_g2_table_build(Engine* e, hipStream_t st, const G2A* hi, size_t half) {
    constexpr int M = 4;
    const size_t qstride = (half + 63) & ~(size_t)63;
    int32_t rc;
    e->g2tab_hi = nullptr;
    if ((rc = e->fold_mult.reserve((size_t)M * half * sizeof(G2A))) || (rc = e->fold_tab.reserve((size_t)4 * M * G2A_CHUNKS * qstride * sizeof(uint4))) ||
        (rc = e->fold_jac2.reserve((size_t)(M - 1) * half * sizeof(G2J))) || (rc = e->fix_flags.reserve(4 * half + 16))) return rc;
    e->tab_owner = nullptr;                                     // whatever round-0 tables were there are overwritten
    G2A* mult = e->fold_mult.as<G2A>();
    HIPCHK(hipMemcpyAsync(mult, hi, half * sizeof(G2A), hipMemcpyDeviceToDevice, st));
    if (!e->sw.no_fq) {
        hipLaunchKernelGGL(k_odd_multiples_q, dim3(nblk(half, 64)), dim3(64), 0, st, hi, (uint32_t)half, M, e->fold_jac2.as<G2J>(), e->fix_flags.as<uint8_t>());
#if !defined(RIPP_INLINE_FALLBACK)
        hipLaunchKernelGGL(k_odd_multiples_fix, dim3(FIX_GRID), dim3(64), 0, st, hi, (uint32_t)half, M, e->fold_jac2.as<G2J>(), e->fix_flags.as<uint8_t>());
#endif
    }
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_odd_multiples<Fp2>), dim3(nblk(half, 64)), dim3(64), 0, st, hi, (uint32_t)half, M, e->fold_jac2.as<G2J>());
    HIPCHK(hipGetLastError());
    if ((rc = e->normalize_dev<Fp2>(e->fold_jac2.as<G2J>(), (size_t)(M - 1) * half, mult + half, st)) != RIPP_OK) return rc;
    hipLaunchKernelGGL(k_g2_tab_images, dim3(nblk(half, 64), M), dim3(64), 0, st, mult, (uint32_t)half, M, e->fold_tab.as<uint4>(), qstride);
    HIPCHK(hipGetLastError());
    e->g2tab_hi = hi; e->g2tab_half = half;
    return RIPP_OK;
}
int32_t fold_g2_table(Engine* e, hipStream_t st, const G2A* hi, const G2A* lo, size_t half, const Fr& s, DevBuf& jac) {
    constexpr int M = 4;
    const size_t qstride = (half + 63) & ~(size_t)63;
    int32_t rc;
    const bool built = e->g2tab_hi == hi && e->g2tab_half == half && e->tab_owner == nullptr;      // (built on this stream: ordered before the fold)
    if (!built && (rc = fold_g2_table_build(e, st, hi, half))) return rc;
    e->g2tab_hi = nullptr;
    if ((rc = jac.reserve(half * sizeof(G2J))) || (rc = e->fix_flags.reserve(4 * half + 16))) return rc;
    if (!e->sw.no_fq && half >= e->fq_min)
    {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fold_g2_tab_q<GlsDigits, 4>), dim3(nblk(half, 64)), dim3(64), 0, st, e->fold_tab.as<uint4>(), qstride, M, lo, (uint32_t)half, gls_wnaf(s, 4), jac.as<G2J>(), e->fix_flags.as<uint8_t>());
#if !defined(RIPP_INLINE_FALLBACK)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fold_g2_tab_fix<GlsDigits, 4>), dim3(FIX_GRID), dim3(64), 0, st, e->fold_tab.as<uint4>(), qstride, M, lo, (uint32_t)half, gls_wnaf(s, 4), jac.as<G2J>(), e->fix_flags.as<uint8_t>());
#endif
    }
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fold_g2_tab<GlsDigits, 4>), dim3(nblk(half, 64)), dim3(64), 0, st, e->fold_tab.as<uint4>(), qstride, M, lo, (uint32_t)half, gls_wnaf(s, 4), jac.as<G2J>());
    HIPCHK(hipGetLastError());
    return RIPP_OK;
}

// Rounds whose G2 fold takes the in-round table form (fold_g2_table: 2^15 <= half, no round-0 tables): the tables depend on the vector only, so they are
// enqueued as soon as the round's pairing products have left the device -- the ~0.9 ms in which the host computes the two final exponentiations and the
// challenge would otherwise be idle device time on the proof's critical path.  Mirrors job_fold's choice of form; a wrong guess costs the table work only.
int32_t job_prebuild_g2_tables(Engine* e, ripp_sipp_job* j) {
    const size_t half = j->len / 2;
    e->g2tab_hi = nullptr;
    if (e->sw.no_prebuild || e->sw.no_endo || half < 1) return RIPP_OK;
    const bool vm_form = (half <= e->vm_fold_max || half <= e->vm_joint_max) && !e->sw.no_vm;
    if (vm_form || j->pre_ready || j->pre_vm_ready || (j->tab_ready && e->tab_owner == j) || !fold_g2_table_pays(e, half)) return RIPP_OK;
    const bool xs = xscale_round(e, j, half);               // (challenges of SIPP are 128-bit: fits_128 holds)
    if (j->bs_on && !xs) return RIPP_OK;                    // the un-scaling fold takes no tables
    if (!fold_g2_table_fits(e, half)) return RIPP_OK;      // short of memory: job_fold takes the split form
    const G2A* b = j->b.as<G2A>();
    return fold_g2_table_build(e, e->stream, xs ? b : b + half, half);
}

int32_t job_fold(Engine* e, ripp_sipp_job* j, const Fr& x, bool allow_vm = true, bool async = false) {
    const size_t half = j->len / 2;
    int32_t rc;
    const Fr x_inv = inv(x);                                                        // sipp/src/lib.rs:94
    hipEvent_t t0 = e->ev_t0, t1 = e->ev_t1;
    HIPCHK(hipEventRecord(t0, e->stream));
    if (async) { hipEvent_t fa, fb; HIPCHK(hipEventCreate(&fa)); HIPCHK(hipEventCreate(&fb)); e->ev_fold_async.emplace_back(fa, fb); HIPCHK(hipEventRecord(fa, e->stream)); }
    G1A* a = j->a.as<G1A>(); G2A* b = j->b.as<G2A>();
    const size_t qstride = (half + 63) & ~(size_t)63;
    if ((rc = e->qtab.reserve(4 * G2A_CHUNKS * qstride * sizeof(uint4))) != RIPP_OK) return rc;
    // G1 half on stream2, G2 half on the main stream (small rounds leave most of the chip idle otherwise)
    const bool use_vm = allow_vm && half <= e->vm_fold_max && !e->sw.no_vm;
    const bool mid_vm = allow_vm && !use_vm && half <= e->vm_joint_max && !e->sw.no_vm && !e->sw.no_endo;      // mid-size rounds: one VM group per element
    if ((rc = e->vm_flag.reserve(sizeof(uint32_t))) != RIPP_OK) return rc;      // (kernel argument of the VM folds; they use the complete addition law and report nothing)
    const bool pre = j->pre_ready && fits_128(x); j->pre_ready = false;      // second bases prepared in the hash window (job_precompute_round0)
    const bool pre_vm = j->pre_vm_ready && fits_128(x) && allow_vm; j->pre_vm_ready = false;   // ... or on the VM during this round's host phase
    const hipStream_t g1s = e->stream2;
    HIPCHK(hipEventRecord(e->ev_fork, e->stream));
    HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
    const bool tab = j->tab_ready && e->tab_owner == j && fits_128(x); j->tab_ready = false;        // ... with the odd-multiple tables of four bases
    const int tabW = tab_width(j->tab_M);
    const G1A* tab1 = e->fold_tab1.as<G1A>() + (j->tab_fused ? half / 2 : 0);      // three-quarter tables start at A1: a_r is q elements in
    if (tab && !e->sw.no_fq && half >= e->fq_min)
        hipLaunchKernelGGL(k_fold_g1_tab_q, dim3(nblk(half, 256)), dim3(256), 0, g1s, tab1, j->tab_cnt, j->tab_M, a, (uint32_t)half, split32_wnaf(x, tabW), j->jac1.as<G1J>());
    else if (tab)
        hipLaunchKernelGGL(k_fold_g1_tab, dim3(nblk(half, 256)), dim3(256), 0, g1s, tab1, j->tab_cnt, j->tab_M, a, (uint32_t)half, split32_wnaf(x, tabW), j->jac1.as<G1J>());
    else if (pre)
        hipLaunchKernelGGL(k_fold_g1_two, dim3(nblk(half, 256)), dim3(256), 0, g1s, a + half, j->a_pow.as<G1A>(), a, (uint32_t)half, split64_digits(x), j->jac1.as<G1J>());
    else if (pre_vm) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_vm_fold_split2<Fp>), dim3(nblk(half, 4 * VM_EPW), 2), dim3(256), 4 * VM_EPW * VmCurve<Fp>::SLOTS * sizeof(VmSlot), g1s, a + half, j->a_pow_h.as<G1J>(), (uint32_t)half, split_digits_g1(x), 1, j->parts1.as<G1J>());
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_vm_combine<Fp>), dim3(nblk(half, 4 * VM_EPW)), dim3(256), 4 * VM_EPW * VmCurve<Fp>::SLOTS * sizeof(VmSlot), g1s, j->parts1.as<G1J>(), 2, a, (uint32_t)half, j->jac1.as<G1J>());
    }
    else if (use_vm || mid_vm)
        hipLaunchKernelGGL(k_vm_fold_g1, dim3(nblk(half, 4 * VM_EPW)), dim3(256), 4 * VM_EPW * VM_G1_SLOTS * sizeof(VmSlot), g1s, a + half, a, (uint32_t)half, naf_digits(x), j->jac1.as<G1J>(), e->vm_flag.as<uint32_t>());
    else if (!e->sw.no_fq && half >= e->fq_min_g1)
        hipLaunchKernelGGL(k_fold_g1_naf_q, dim3(nblk(half, 256)), dim3(256), 0, g1s, a + half, a, (uint32_t)half, naf_digits(x), j->jac1.as<G1J>());
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fold_affine_naf<Fp>), dim3(nblk(half, 256)), dim3(256), 0, g1s, a + half, a, (uint32_t)half, naf_digits(x), j->jac1.as<G1J>());
    HIPCHK(hipGetLastError());
    if ((rc = e->normalize_dev<Fp>(j->jac1.as<G1J>(), half, j->a_next.as<G1A>(), g1s)) != RIPP_OK) return rc;
    HIPCHK(hipEventRecord(e->ev_join, g1s));
    // ---- G2 side.  xs: this round folds the x-scaled vector, bt' = x bt_l + bt_r (x multiplies the LOW half, the high half is the addend);
    //      unscale: the vector is scaled by bs but this round is too small for the table fold: b' = (1/bs) bt_l + (1/(bs x)) bt_r in one pass
    const bool xs = xscale_round(e, j, half) && fits_128(x) && (!tab || j->tab_on_lo);
    const bool unscale = j->bs_on && !xs;
    if (tab && j->tab_on_lo != xs) return (set_err("fold tables were built for the other half"), RIPP_ERR_ARG);
    const G2A* g2_hi = xs ? b : b + half; const G2A* g2_lo = xs ? b + half : b;
    const Fr g2_s = xs ? x : x_inv;
    if (unscale) {
        // two full-width scalars on two bases: the 4-lane GLS form on each (8 lanes per element, one dependent chain of 65 doublings), then one sum
        const Fr s_inv = inv(j->bs), sx_inv = mul(s_inv, x_inv);
        if (mid_vm) {                                   // one VM group per element walks all 8 strings: +~1 ms over an ordinary fold of this size
            GlsDigits2 d2; d2.a = gls_digits(s_inv); d2.b = gls_digits(sx_inv);
            hipLaunchKernelGGL(k_vm_fold_g2_joint2, dim3(nblk(half, 4 * VM_EPW)), dim3(256), 4 * VM_EPW * VM_G2_SLOTS * sizeof(VmSlot), e->stream, b, b + half, (uint32_t)half, d2, j->jac2.as<G2J>());
        } else {
        if ((rc = e->qtab.reserve(8 * half * sizeof(G2J))) != RIPP_OK || (rc = e->fix_flags.reserve(8 * half + 32)) != RIPP_OK) return rc;
        HIPCHK(hipStreamWaitEvent(e->stream3, e->ev_fork, 0));                        // the two multiplications side by side: they are latency-bound
        for (int t = 0; t < 2; ++t) {
            hipStream_t st2 = t ? e->stream3 : e->stream; const G2A* src = t ? b + half : b; const GlsDigits dg2 = gls_digits(t ? sx_inv : s_inv);
            G2J* parts = e->qtab.as<G2J>() + (size_t)t * 4 * half; uint8_t* flags = e->fix_flags.as<uint8_t>() + (size_t)t * ((4 * half + 15) & ~(size_t)15);
            if (!e->sw.no_fq) {
                hipLaunchKernelGGL(k_fold_g2_gls_split_q, dim3(nblk(half, 64), 4), dim3(64), 0, st2, src, (uint32_t)half, dg2, parts, flags);
#if !defined(RIPP_INLINE_FALLBACK)
                hipLaunchKernelGGL(k_fold_g2_gls_split_fix, dim3(FIX_GRID), dim3(64), 0, st2, src, (uint32_t)half, dg2, parts, flags);
#endif
            } else hipLaunchKernelGGL(k_fold_g2_gls_split, dim3(nblk(half, 64), 4), dim3(64), 0, st2, src, (uint32_t)half, dg2, parts);
        }
        HIPCHK(hipEventRecord(e->ev_join3, e->stream3)); HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join3, 0));
        hipLaunchKernelGGL(k_fold_g2_combine8, dim3(nblk(half, 64)), dim3(64), 0, e->stream, e->qtab.as<G2J>(), (uint32_t)half, j->jac2.as<G2J>());
        }
        j->bs = Fr::one(); j->bs_on = false;
    } else
    if (pre_vm) {
        if (j->pre_vm_side) { HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join3, 0)); j->pre_vm_side = false; }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_vm_fold_split2<Fp2>), dim3(nblk(half, 4 * VM_EPW), 8), dim3(256), 4 * VM_EPW * VmCurve<Fp2>::SLOTS * sizeof(VmSlot), e->stream, b + half, j->b_pow_h.as<G2J>(), (uint32_t)half, split_digits_g2(x_inv), 4, j->parts2.as<G2J>());
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_vm_combine<Fp2>), dim3(nblk(half, 4 * VM_EPW)), dim3(256), 4 * VM_EPW * VmCurve<Fp2>::SLOTS * sizeof(VmSlot), e->stream, j->parts2.as<G2J>(), 8, b, (uint32_t)half, j->jac2.as<G2J>());
    } else
    if (tab && !e->sw.no_fq && half >= e->fq_min) {
        if ((rc = e->fix_flags.reserve(4 * half + 16))) return rc;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fold_g2_tab_q<Wnaf16, 16>), dim3(nblk(half, 64)), dim3(64), 0, e->stream, e->fold_tab.as<uint4>(), j->tab2_stride, j->tab_M, g2_lo, (uint32_t)half, gls16_wnaf(g2_s, tabW), j->jac2.as<G2J>(), e->fix_flags.as<uint8_t>());
#if !defined(RIPP_INLINE_FALLBACK)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fold_g2_tab_fix<Wnaf16, 16>), dim3(FIX_GRID), dim3(64), 0, e->stream, e->fold_tab.as<uint4>(), j->tab2_stride, j->tab_M, g2_lo, (uint32_t)half, gls16_wnaf(g2_s, tabW), j->jac2.as<G2J>(), e->fix_flags.as<uint8_t>());
#endif
    } else
    if (tab) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fold_g2_tab<Wnaf16, 16>), dim3(nblk(half, 64)), dim3(64), 0, e->stream, e->fold_tab.as<uint4>(), j->tab2_stride, j->tab_M, g2_lo, (uint32_t)half, gls16_wnaf(g2_s, tabW), j->jac2.as<G2J>());
    } else
    if (pre) {
        if ((rc = e->qtab.reserve(8 * G2A_CHUNKS * qstride * sizeof(uint4))) != RIPP_OK) return rc;
        hipLaunchKernelGGL(k_fold_g2_gls8, dim3(nblk(half, 64)), dim3(64), 0, e->stream, b + half, j->b_pow.as<G2A>(), b, (uint32_t)half, gls8_digits(x_inv), e->qtab.as<uint4>(), qstride, j->jac2.as<G2J>());
    } else
    if (mid_vm) {
        hipLaunchKernelGGL(k_vm_fold_g2_joint, dim3(nblk(half, 4 * VM_EPW)), dim3(256), 4 * VM_EPW * VM_G2_SLOTS * sizeof(VmSlot), e->stream, g2_hi, g2_lo, (uint32_t)half, gls_digits(g2_s), j->jac2.as<G2J>());
    } else
    if (use_vm) {
        if ((rc = e->qtab.reserve(std::max<size_t>(4 * G2A_CHUNKS * qstride * sizeof(uint4), 4 * half * sizeof(G2J)))) != RIPP_OK) return rc;
        hipLaunchKernelGGL(k_vm_fold_g2_split, dim3(nblk(half, 4 * VM_EPW), 4), dim3(256), 4 * VM_EPW * VM_G2_SLOTS * sizeof(VmSlot), e->stream, b + half, (uint32_t)half, gls_digits(x_inv), e->qtab.as<G2J>(), e->vm_flag.as<uint32_t>());
        hipLaunchKernelGGL(k_vm_combine_g2, dim3(nblk(half, 4 * VM_EPW)), dim3(256), 4 * VM_EPW * VM_G2_SLOTS * sizeof(VmSlot), e->stream, e->qtab.as<G2J>(), b, (uint32_t)half, j->jac2.as<G2J>());
    } else
    if (e->sw.no_endo) {               // no psi on this build / switch: the 255-bit NAF fold (x^-1 is full width)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fold_affine_naf<Fp2>), dim3(nblk(half, 256)), dim3(256), 0, e->stream, b + half, b, (uint32_t)half, naf_digits(x_inv), j->jac2.as<G2J>());
    } else
    if (half <= e->gls_split_max || (fold_g2_table_pays(e, half) && !(e->g2tab_hi && e->g2tab_half == half) && !fold_g2_table_fits(e, half) && ((e->mem_tier |= 16), true))) {     // latency-bound round: 4 lanes per element (also: a table round whose tables the device cannot hold, mem_tier + 16)
        if ((rc = e->qtab.reserve(std::max<size_t>(4 * G2A_CHUNKS * qstride * sizeof(uint4), 4 * half * sizeof(G2J)))) != RIPP_OK) return rc;
        if (!e->sw.no_fq) {
            if ((rc = e->fix_flags.reserve(4 * half + 16))) return rc;
            hipLaunchKernelGGL(k_fold_g2_gls_split_q, dim3(nblk(half, 64), 4), dim3(64), 0, e->stream, g2_hi, (uint32_t)half, gls_digits(g2_s), e->qtab.as<G2J>(), e->fix_flags.as<uint8_t>());
#if !defined(RIPP_INLINE_FALLBACK)
            hipLaunchKernelGGL(k_fold_g2_gls_split_fix, dim3(FIX_GRID), dim3(64), 0, e->stream, g2_hi, (uint32_t)half, gls_digits(g2_s), e->qtab.as<G2J>(), e->fix_flags.as<uint8_t>());
#endif
        }
        else hipLaunchKernelGGL(k_fold_g2_gls_split, dim3(nblk(half, 64), 4), dim3(64), 0, e->stream, g2_hi, (uint32_t)half, gls_digits(g2_s), e->qtab.as<G2J>());
        hipLaunchKernelGGL(k_fold_g2_combine, dim3(nblk(half, 64)), dim3(64), 0, e->stream, e->qtab.as<G2J>(), g2_lo, (uint32_t)half, j->jac2.as<G2J>());
    } else if (fold_g2_table_pays(e, half)) {
        if ((rc = fold_g2_table(e, e->stream, g2_hi, g2_lo, half, g2_s, j->jac2)) != RIPP_OK) return rc;
    } else {
        hipLaunchKernelGGL(k_fold_g2_gls, dim3(nblk(half, 64)), dim3(64), 0, e->stream, b + half, b, (uint32_t)half, gls_digits(x_inv), e->qtab.as<uint4>(), qstride, j->jac2.as<G2J>());
    }
    HIPCHK(hipGetLastError());
    if (xs) { j->bs = j->bs_on ? mul(j->bs, x) : x; j->bs_on = true; }          // the new vector is (bs x) b'
    if ((rc = e->normalize_dev<Fp2>(j->jac2.as<G2J>(), half, j->b_next.as<G2A>())) != RIPP_OK) return rc;
    HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join, 0));
    HIPCHK(hipEventRecord(t1, e->stream));
    if (!async) {                 // pipelined tail / pre-evaluated rounds (async): everything that follows is ordered behind the fold on the engine's streams
        if ((rc = e->sync()) != RIPP_OK) return rc;
        float ms = 0; (void)hipEventElapsedTime(&ms, t0, t1); e->stats.fold_ms += ms;
    } else HIPCHK(hipEventRecord(e->ev_fold_async.back().second, e->stream));
    std::swap(j->a, j->a_next); std::swap(j->b, j->b_next);
    j->len = half;
    return RIPP_OK;
}

// Rounds 0 and 1 folded in ONE pass over the three-quarter tables of job_precompute_round0 (fq_curve.hpp / fq_curve2.hpp have the algebra):
//     a''_i  = A0_i + x0 A2_i + x1 A1_i + (x0 x1) A3_i,        bt''_i = (x0 x1) B0_i + x0 B1_i + x1 B2_i + B3_i        (i < q = len / 4)
// -- the vectors the two folds of job_fold produce one after the other (bt'' = x0 x1 b'': the G2 vector stays x-scaled, bs = x0 x1).  Called with the
// job's vectors still UNFOLDED (j->len = the round-0 length) once both challenges are known; leaves j->len = len / 4.
int32_t job_fold_fused(Engine* e, ripp_sipp_job* j, const Fr& x0, const Fr& x1, bool async) {
    const size_t q = j->len / 4;
    int32_t rc;
    if (!(j->tab_ready && j->tab_fused && e->tab_owner == j && j->tab_on_lo && !j->bs_on && fits_128(x0) && fits_128(x1))) { set_err("fused fold: the three-quarter tables of this job are not in place"); return RIPP_ERR_ARG; }
    j->tab_ready = false; j->pre_ready = false; j->pre_vm_ready = false;
    hipEvent_t t0 = e->ev_t0, t1 = e->ev_t1;
    HIPCHK(hipEventRecord(t0, e->stream));
    if (async) { hipEvent_t fa, fb; HIPCHK(hipEventCreate(&fa)); HIPCHK(hipEventCreate(&fb)); e->ev_fold_async.emplace_back(fa, fb); HIPCHK(hipEventRecord(fa, e->stream)); }
    if ((rc = j->jac1.reserve(q * sizeof(G1J))) || (rc = j->jac2.reserve(q * sizeof(G2J))) || (rc = e->fix_flags.reserve(2 * ((q + 15) & ~(size_t)15) + 32))) return rc;
    const int W = tab_width(j->tab_M);
    G1A* a = j->a.as<G1A>(); G2A* b = j->b.as<G2A>();
    uint8_t* flag1 = e->fix_flags.as<uint8_t>(); uint8_t* flag2 = flag1 + ((q + 15) & ~(size_t)15) + 16;
    HIPCHK(hipEventRecord(e->ev_fork, e->stream));
    HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
    // G1 on stream2
    const hipStream_t g1s = e->stream2;
    const WnafG1x4 d1 = fused_digits_g1(x0, x1, W);
    hipLaunchKernelGGL(k_fold_g1_fused_q, dim3(nblk(q, 256)), dim3(256), 0, g1s, e->fold_tab1.as<G1A>(), j->tab_cnt, j->tab_M, a, (uint32_t)q, d1, j->jac1.as<G1J>(), flag1);
    hipLaunchKernelGGL(k_fold_g1_fused_fix, dim3(FIX_GRID), dim3(64), 0, g1s, e->fold_tab1.as<G1A>(), j->tab_cnt, j->tab_M, a, (uint32_t)q, d1, j->jac1.as<G1J>(), flag1);
    HIPCHK(hipGetLastError());
    if ((rc = e->normalize_dev<Fp>(j->jac1.as<G1J>(), q, j->a_next.as<G1A>(), g1s)) != RIPP_OK) return rc;
    HIPCHK(hipEventRecord(e->ev_join, g1s));
    // G2 on the main stream
    const Wnaf16x3 d2 = fused_digits_g2(x0, x1, W);
    hipLaunchKernelGGL(k_fold_g2_fused_q, dim3(nblk(q, 64)), dim3(64), 0, e->stream, e->fold_tab.as<uint4>(), j->tab2_stride, j->tab_M, b + 3 * q, (uint32_t)q, d2, j->jac2.as<G2J>(), flag2);
    hipLaunchKernelGGL(k_fold_g2_fused_fix, dim3(FIX_GRID), dim3(64), 0, e->stream, e->fold_tab.as<uint4>(), j->tab2_stride, j->tab_M, b + 3 * q, (uint32_t)q, d2, j->jac2.as<G2J>(), flag2);
    HIPCHK(hipGetLastError());
    j->bs = mul(x0, x1); j->bs_on = true;                                        // the new vector is (x0 x1) b''
    if ((rc = e->normalize_dev<Fp2>(j->jac2.as<G2J>(), q, j->b_next.as<G2A>())) != RIPP_OK) return rc;
    HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join, 0));
    HIPCHK(hipEventRecord(t1, e->stream));
    if (!async) {
        if ((rc = e->sync()) != RIPP_OK) return rc;
        float ms = 0; (void)hipEventElapsedTime(&ms, t0, t1); e->stats.fold_ms += ms;
    } else HIPCHK(hipEventRecord(e->ev_fold_async.back().second, e->stream));
    std::swap(j->a, j->a_next); std::swap(j->b, j->b_next);
    j->len = q;
    return RIPP_OK;
}

void job_start_hash(ripp_sipp_job* j, const Fp12& value) {
    if (j->hash_prestarted) { j->hash_prestarted = false; return; }      // ripp_sipp_prove started it before the upload
    if (j->hash_thread.joinable()) j->hash_thread.join();
    j->digest_ready = false;
    const G1A* pa = j->ha_ext ? j->ha_ext : j->ha.data(); const G2A* pb = j->ha_ext ? j->hb_ext : j->hb.data(); const Fr* pr = j->ha_ext ? j->hr_ext : j->hr.data();
    const size_t pn = j->ha_ext ? j->hash_n : j->ha.size();
    j->hash_done = 0; j->hash_total = (uint64_t)pn * (96 + 192 + 32); j->hash_t0 = now_ms();
    j->hash_thread = std::thread([j, value, pa, pb, pr, pn]() { statement_digest(pa, pb, pr, pn, value, j->digest, &j->hash_done); j->digest_ready = true; });
}

}  // namespace

// =============================================================================================== C ABI
extern "C" {

#define API __attribute__((visibility("default")))
#define LOCK std::lock_guard<std::mutex> lk_(g_mu)
#define ENGINE Engine* e; { int32_t rc_ = get_engine(&e); if (rc_ != RIPP_OK) return rc_; }

API const char* ripp_last_error(void) { return g_err.c_str(); }
static_assert(sizeof(ripp_stats) == 24 * 8, "ripp_stats changed: bump RIPP_ABI_VERSION (include/ripp_hip.h) and the bindings (ripp_amd/_lib.py, rust/ripp-hip/src/ffi.rs)");
API int32_t ripp_abi_version(void) { return RIPP_ABI_VERSION; }
API size_t ripp_stats_size(void) { return sizeof(ripp_stats); }
API void ripp_statement_hash_times(double* hash_ms, double* wait_ms) { if (hash_ms) *hash_ms = g_digest_hash_ms; if (wait_ms) *wait_ms = g_digest_wait_ms; }
API int32_t ripp_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
extern "C++" { static void vec_caches_release(); }      // tipa_api.inc: the parked device buffers of the GIPA / TIPA vector sets (they belong to the engine's device)
API int32_t ripp_init(int32_t dev) {
    LOCK;
    if (g_engine) {
        if (g_engine->device == dev) return RIPP_OK;
        // re-binding to another device would free the engine's streams under live job / SRS handles: refuse instead
        if (g_live_handles.load() > 0) { set_err("ripp_init: " + std::to_string(g_live_handles.load()) + " job / SRS handle(s) are alive on device " + std::to_string(g_engine->device) + "; destroy them before re-binding the engine"); return RIPP_ERR_ARG; }
        vec_caches_release();
        g_engine->destroy(); delete g_engine; g_engine = nullptr;
    }
    Engine* e = new Engine(); int32_t rc = e->init(dev);
    if (rc != RIPP_OK) { delete e; return rc; }
    g_engine = e; return RIPP_OK;
}
API void ripp_shutdown(void) {
    LOCK; if (!g_engine) return;
    // job / SRS / vector handles hold device memory and refer to this engine's streams and tables: tearing it down under them would let the
    // next call create a fresh engine (possibly on another device) that those handles then run on.  Refuse, like ripp_init does.
    if (g_live_handles.load() > 0) { set_err("ripp_shutdown: " + std::to_string(g_live_handles.load()) + " job / SRS / vector handle(s) are still alive; destroy them first"); fprintf(stderr, "[ripp] %s\n", g_err.c_str()); return; }
    vec_caches_release();
    g_engine->destroy(); delete g_engine; g_engine = nullptr;
}
// frees the engine's grow-only scratch (line buffer, fold tables -- ~19 GB after an n = 2^20 proof --, MSM scratch): the next call re-allocates
// what it needs.  Job and SRS handles keep their own buffers.
API int32_t ripp_release_scratch(void) {
    LOCK; if (!g_engine) return RIPP_OK;
    Engine* e = g_engine;
    if (hipSetDevice(e->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { set_err("ripp_release_scratch: device synchronisation failed"); return RIPP_ERR_DEVICE; }
    for (DevBuf* b : {&e->lines, &e->partA, &e->partB, &e->jacG1, &e->jacG2, &e->tmpA, &e->tmpB, &e->tmpR, &e->affG1, &e->affG2, &e->qtab, &e->scale_tab, &e->fold_tab1, &e->fold_mult, &e->fold_tab, &e->fold_jac1, &e->fold_jac2, &e->fix_flags, &e->scale_flags}) b->release();
    e->msm_scratch[0].release(); e->msm_scratch[1].release(); e->kzg_q[0].release(); e->kzg_q[1].release(); e->kzg_bases[0].release(); e->kzg_bases[1].release();
    for (PinBuf& pb : e->stage) pb.release();             // pinned host staging (up to 2 x 32 MB after a verifier call at n = 2^20)
    e->tab_owner = nullptr; e->g2tab_hi = nullptr; e->job_cache.release(); vec_caches_release();
    if (e->aux) { e->aux->destroy(); delete e->aux; e->aux = nullptr; }
    for (Engine* p : e->peers) { (void)hipSetDevice(p->device); p->destroy(); delete p; }
    if (!e->peers.empty()) { e->peers.clear(); (void)hipSetDevice(e->device); }
    return RIPP_OK;
}

// device memory the library holds right now through its own buffers (scratch, tables, jobs, SRS / vector handles): what ripp_config.mem_cap_bytes bounds
API size_t ripp_device_bytes(void) { return g_dev_bytes.load(std::memory_order_relaxed); }

// ---- normalisation / scaling / folds on host slices ---------------------------------------------------------
API int32_t ripp_normalize_g1(const ripp_g1j* in, size_t n, ripp_g1a* out) {
    LOCK; ENGINE; if (n == 0) return RIPP_OK; if (!in || !out) return RIPP_ERR_ARG;
    G1J* d; int32_t rc = upload<G1J>(e, e->jacG1, in, n, &d); if (rc) return rc;
    if ((rc = e->affG1.reserve(n * sizeof(G1A)))) return rc;
    if ((rc = e->normalize_dev<Fp>(d, n, e->affG1.as<G1A>()))) return rc;
    HIPCHK(hipMemcpyAsync(out, e->affG1.p, n * sizeof(G1A), hipMemcpyDeviceToHost, e->stream));
    return e->sync();
}
API int32_t ripp_normalize_g2(const ripp_g2j* in, size_t n, ripp_g2a* out) {
    LOCK; ENGINE; if (n == 0) return RIPP_OK; if (!in || !out) return RIPP_ERR_ARG;
    G2J* d; int32_t rc = upload<G2J>(e, e->jacG2, in, n, &d); if (rc) return rc;
    if ((rc = e->affG2.reserve(n * sizeof(G2A)))) return rc;
    if ((rc = e->normalize_dev<Fp2>(d, n, e->affG2.as<G2A>()))) return rc;
    HIPCHK(hipMemcpyAsync(out, e->affG2.p, n * sizeof(G2A), hipMemcpyDeviceToHost, e->stream));
    return e->sync();
}
API int32_t ripp_scale_g1_a(const ripp_g1a* a, const ripp_fr* r, size_t n, ripp_g1a* out) {
    LOCK; ENGINE; if (n == 0) return RIPP_OK; if (!a || !r || !out) return RIPP_ERR_ARG;
    G1A* da; Fr* dr; int32_t rc;
    if ((rc = upload<G1A>(e, e->tmpA, a, n, &da))) return rc;
    if ((rc = upload<Fr>(e, e->tmpR, r, n, &dr))) return rc;
    if ((rc = e->jacG1.reserve(n * sizeof(G1J)))) return rc;
    if ((rc = e->affG1.reserve(n * sizeof(G1A)))) return rc;
    if ((rc = e->scale_g1_dev(da, 1u, dr, n, e->jacG1.as<G1J>()))) return rc;
    if ((rc = e->normalize_dev<Fp>(e->jacG1.as<G1J>(), n, e->affG1.as<G1A>()))) return rc;
    HIPCHK(hipMemcpyAsync(out, e->affG1.p, n * sizeof(G1A), hipMemcpyDeviceToHost, e->stream));
    return e->sync();
}

extern "C++" {
template <class F, bool JAC_IN, bool AFF_OUT, class IN, class OUT>
static int32_t fold_impl(const IN* hi, const IN* lo, size_t half, const ripp_fr* s, OUT* out) {
    LOCK; ENGINE; if (half == 0) return RIPP_OK; if (!hi || !lo || !s || !out) return RIPP_ERR_ARG;
    Fr sm; std::memcpy(&sm, s, sizeof(Fr));
    const ScalarBits sb = scalar_bits(sm);
    int32_t rc;
    DevBuf& jac = std::is_same<F, Fp>::value ? e->jacG1 : e->jacG2;
    DevBuf& aff = std::is_same<F, Fp>::value ? e->affG1 : e->affG2;
    if ((rc = jac.reserve(half * sizeof(Jac<F>)))) return rc;
    if (JAC_IN) {
        Jac<F>*dh, *dl;
        if ((rc = upload<Jac<F>>(e, e->tmpA, hi, half, &dh))) return rc;
        if ((rc = upload<Jac<F>>(e, e->tmpB, lo, half, &dl))) return rc;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fold_jac<F>), dim3(nblk(half, 64)), dim3(64), 0, e->stream, dh, dl, (uint32_t)half, sb, jac.as<Jac<F>>());
    } else {
        Affine<F>*dh, *dl;
        if ((rc = upload<Affine<F>>(e, e->tmpA, hi, half, &dh))) return rc;
        if ((rc = upload<Affine<F>>(e, e->tmpB, lo, half, &dl))) return rc;
        if (std::is_same<F, Fp2>::value && !e->sw.no_endo) {
            const size_t qstride = (half + 63) & ~(size_t)63;
            if ((rc = e->qtab.reserve(4 * G2A_CHUNKS * qstride * sizeof(uint4)))) return rc;
            hipLaunchKernelGGL(k_fold_g2_gls, dim3(nblk(half, 64)), dim3(64), 0, e->stream, reinterpret_cast<const G2A*>(dh), reinterpret_cast<const G2A*>(dl), (uint32_t)half,
                               gls_digits(sm), e->qtab.as<uint4>(), qstride, reinterpret_cast<G2J*>(jac.p));
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fold_affine_naf<F>), dim3(nblk(half, 256)), dim3(256), 0, e->stream, dh, dl, (uint32_t)half, naf_digits(sm), jac.as<Jac<F>>());
        }
    }
    HIPCHK(hipGetLastError());
    if (AFF_OUT) {
        if ((rc = aff.reserve(half * sizeof(Affine<F>)))) return rc;
        if ((rc = e->normalize_dev<F>(jac.as<Jac<F>>(), half, aff.as<Affine<F>>()))) return rc;
        HIPCHK(hipMemcpyAsync(out, aff.p, half * sizeof(Affine<F>), hipMemcpyDeviceToHost, e->stream));
    } else {
        HIPCHK(hipMemcpyAsync(out, jac.p, half * sizeof(Jac<F>), hipMemcpyDeviceToHost, e->stream));
    }
    return e->sync();
}
}  // extern "C++"
API int32_t ripp_fold_g1_a(const ripp_g1a* hi, const ripp_g1a* lo, size_t half, const ripp_fr* s, ripp_g1a* out) { return fold_impl<Fp, false, true>(hi, lo, half, s, out); }
API int32_t ripp_fold_g2_a(const ripp_g2a* hi, const ripp_g2a* lo, size_t half, const ripp_fr* s, ripp_g2a* out) { return fold_impl<Fp2, false, true>(hi, lo, half, s, out); }
API int32_t ripp_fold_g1_j(const ripp_g1j* hi, const ripp_g1j* lo, size_t half, const ripp_fr* s, ripp_g1j* out) { return fold_impl<Fp, true, false>(hi, lo, half, s, out); }
API int32_t ripp_fold_g2_j(const ripp_g2j* hi, const ripp_g2j* lo, size_t half, const ripp_fr* s, ripp_g2j* out) { return fold_impl<Fp2, true, false>(hi, lo, half, s, out); }

// scalar-vector fold of GIPA (gipa.rs:270-274 with Message = Fr): out[i] = hi[i] * s + lo[i]
API int32_t ripp_fold_fr(const ripp_fr* hi, const ripp_fr* lo, size_t half, const ripp_fr* s, ripp_fr* out) {
    LOCK; ENGINE; if (half == 0) return RIPP_OK; if (!hi || !lo || !s || !out) return RIPP_ERR_ARG;
    Fr sm; std::memcpy(&sm, s, sizeof sm);
    Fr *dh, *dl; int32_t rc;
    if ((rc = upload<Fr>(e, e->tmpA, hi, half, &dh)) || (rc = upload<Fr>(e, e->tmpB, lo, half, &dl)) || (rc = e->tmpR.reserve(half * sizeof(Fr)))) return rc;
    hipLaunchKernelGGL(k_fold_fr, dim3(nblk(half, 256)), dim3(256), 0, e->stream, dh, dl, (uint32_t)half, sm, e->tmpR.as<Fr>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, e->tmpR.p, half * sizeof(Fr), hipMemcpyDeviceToHost, e->stream));
    return e->sync();
}

// ---- in-process multi-device dispatch (ripp_config.n_devices) --------------------------------------------------------------------------------
// The reference's traits are static, stateless calls of ONE process (inner_products/src/lib.rs:45-48), so an unmodified caller of PairingInnerProduct /
// MultiexponentiationInnerProduct can only ever reach the devices this process drives.  With n_devices = D > 1 a host-slice call is cut into D contiguous
// index ranges; range d runs on its own Engine (device bound + d; streams, scratch, line buffer of its own) under its own host thread, the D partial results
// -- 68 per-step Fp12 products, or one group element -- come back by plain D2H copies, and the host combines them and runs the ONE final exponentiation.
// No communicator, no RCCL: the host needs the value anyway.  The result is the single-device call's value (a projective MSM result
// is another representative of the same point).  UNMEASURED on a multi-GPU node; tested on one GPU with RIPP_VIRTUAL_DEVICES.
extern "C++" {
static std::atomic<int32_t> g_last_slots{1};
struct DevSlot { Engine* e; size_t off, cnt; };
static int32_t device_slots(Engine* e, size_t n, size_t min_units, std::vector<DevSlot>* out) {
    size_t D = e->n_devices_cfg > 1 ? e->n_devices_cfg : 1;
    if (min_units && D > n / min_units) D = std::max<size_t>(1, n / min_units);
    out->clear();
    if (D > 1) {
        int count = 0; (void)hipGetDeviceCount(&count);
        for (size_t d = 1; d < D; ++d) {
            const int phys = e->virtual_devices ? e->device : e->device + (int)d;
            if (phys >= count) { set_err("n_devices = " + std::to_string(e->n_devices_cfg) + ": device " + std::to_string(phys) + " does not exist (" + std::to_string(count) + " visible)"); (void)hipSetDevice(e->device); return RIPP_ERR_ARG; }
            if (e->peers.size() < d) {
                Engine* p = new Engine(); const int32_t rc = p->init(phys);
                (void)hipSetDevice(e->device);
                if (rc != RIPP_OK) { delete p; return rc; }
                e->peers.push_back(p);
            } else if (e->peers[d - 1]->device != phys) { set_err("n_devices: the device mapping changed under live peer engines (ripp_release_scratch first)"); return RIPP_ERR_ARG; }
            e->peers[d - 1]->refresh_switches(); e->peers[d - 1]->stats = ripp_stats{};
        }
    }
    for (size_t d = 0; d < D; ++d) { const size_t lo = n * d / D, hi = n * (d + 1) / D; out->push_back({d == 0 ? e : e->peers[d - 1], lo, hi - lo}); }
    g_last_slots = (int32_t)D;
    return RIPP_OK;
}
// fn(engine, offset, count, slot) on every slot: slot 0 on the calling thread, the others on threads of their own (HIP's current device is per thread)
template <class FN> static int32_t run_on_devices(Engine* e, const std::vector<DevSlot>& sl, FN fn) {
    if (sl.size() == 1) return fn(sl[0].e, sl[0].off, sl[0].cnt, 0);
    std::vector<int32_t> rcs(sl.size(), RIPP_OK); std::vector<std::string> errs(sl.size());
    std::vector<std::thread> th;
    for (size_t d = 1; d < sl.size(); ++d)
        th.emplace_back([&, d]() {
            if (hipSetDevice(sl[d].e->device) != hipSuccess) { rcs[d] = RIPP_ERR_DEVICE; errs[d] = "hipSetDevice failed"; return; }
            t_engine = sl[d].e;
            rcs[d] = fn(sl[d].e, sl[d].off, sl[d].cnt, (int)d);
            if (rcs[d]) errs[d] = g_err;
        });
    rcs[0] = fn(sl[0].e, sl[0].off, sl[0].cnt, 0); if (rcs[0]) errs[0] = g_err;
    for (std::thread& t : th) t.join();
    (void)hipSetDevice(e->device);
    for (size_t d = 0; d < sl.size(); ++d) if (rcs[d]) { set_err("device slot " + std::to_string(d) + " of " + std::to_string(sl.size()) + ": " + errs[d]); return rcs[d]; }
    return RIPP_OK;
}
constexpr size_t PAIRS_PER_DEVICE_MIN = 4096, MSM_TERMS_PER_DEVICE_MIN = (size_t)1 << 15;
}  // extern "C++"
API int32_t ripp_device_slots_used(void) { return g_last_slots.load(); }

// ---- pairing products ------------------------------------------------------------------------------------------
static int32_t pairing_product_dev(Engine* e, const G1A* da, const G2A* db, size_t n, ripp_gt* out) {
    Fp12 rows[N_LINES];
    const G1A* as[1] = {da}; const G2A* bs[1] = {db};
    int32_t rc = e->step_products(as, bs, 1, n, rows); if (rc) return rc;
    Fp12 z; pairing_values(rows, 1, &z);
    std::memcpy(out, &z, sizeof(Fp12));
    e->collect_kernel_stats();
    return RIPP_OK;
}
extern "C++" {
// per-step products of one slot's pairs, then the product over the slots, the 63 squarings and ONE final exponentiation on the host
template <class ROWS> static int32_t pairing_product_slots(Engine* e, size_t n, ripp_gt* out, ROWS rows_of /* (engine, off, cnt, Fp12 rows[68]) */) {
    std::vector<DevSlot> sl; int32_t rc = device_slots(e, n, PAIRS_PER_DEVICE_MIN, &sl); if (rc) return rc;
    std::vector<Fp12> rows(sl.size() * N_LINES);
    if ((rc = run_on_devices(e, sl, [&](Engine* ee, size_t off, size_t cnt, int d) { return rows_of(ee, off, cnt, rows.data() + (size_t)d * N_LINES); }))) return rc;
    for (size_t d = 1; d < sl.size(); ++d) for (int s = 0; s < N_LINES; ++s) rows[s] = mul(rows[s], rows[d * N_LINES + s]);
    Fp12 z; pairing_values(rows.data(), 1, &z);
    std::memcpy(out, &z, sizeof(Fp12));
    e->collect_kernel_stats();
    return RIPP_OK;
}
}
API int32_t ripp_pairing_product_a(const ripp_g1a* a, const ripp_g2a* b, size_t n, ripp_gt* out) {
    LOCK; ENGINE; if (!out || (n && (!a || !b))) return RIPP_ERR_ARG;
    return pairing_product_slots(e, n, out, [&](Engine* ee, size_t off, size_t cnt, Fp12* rows) -> int32_t {
        G1A* da; G2A* db; int32_t rc;
        if ((rc = upload<G1A>(ee, ee->tmpA, a + off, cnt, &da)) || (rc = upload<G2A>(ee, ee->tmpB, b + off, cnt, &db))) return rc;
        const G1A* as[1] = {da}; const G2A* bs[1] = {db};
        return ee->step_products(as, bs, 1, cnt, rows); });
}
API int32_t ripp_pairing_product_j(const ripp_g1j* l, size_t nl, const ripp_g2j* r, size_t nr, ripp_gt* out) {
    if (nl != nr) { set_err("left length, right length: " + std::to_string(nl) + ", " + std::to_string(nr)); return RIPP_ERR_LENGTH; }
    LOCK; ENGINE; if (!out || (nl && (!l || !r))) return RIPP_ERR_ARG;
    return pairing_product_slots(e, nl, out, [&](Engine* ee, size_t off, size_t cnt, Fp12* rows) -> int32_t {
        G1J* dl; G2J* dr; int32_t rc;
        if ((rc = upload<G1J>(ee, ee->jacG1, l + off, cnt, &dl)) || (rc = upload<G2J>(ee, ee->jacG2, r + off, cnt, &dr))) return rc;
        if ((rc = ee->affG1.reserve(std::max<size_t>(cnt, 1) * sizeof(G1A))) || (rc = ee->affG2.reserve(std::max<size_t>(cnt, 1) * sizeof(G2A)))) return rc;
        if ((rc = ee->normalize_dev<Fp>(dl, cnt, ee->affG1.as<G1A>()))) return rc;       // inner_products/src/lib.rs:80-81
        if ((rc = ee->normalize_dev<Fp2>(dr, cnt, ee->affG2.as<G2A>()))) return rc;
        const G1A* as[1] = {ee->affG1.as<G1A>()}; const G2A* bs[1] = {ee->affG2.as<G2A>()};
        return ee->step_products(as, bs, 1, cnt, rows); });
}
// this rank's share of a sharded pairing product: the Miller value of its pairs, BEFORE the final exponentiation.
// prod over ranks of these (ripp_combine_partials), then ONE ripp_final_exp, equals ripp_pairing_product_j of the whole vectors.
API int32_t ripp_pairing_miller_j(const ripp_g1j* l, size_t nl, const ripp_g2j* r, size_t nr, ripp_gt* out) {
    if (nl != nr) { set_err("left length, right length: " + std::to_string(nl) + ", " + std::to_string(nr)); return RIPP_ERR_LENGTH; }
    LOCK; ENGINE; if (!out || (nl && (!l || !r))) return RIPP_ERR_ARG;
    G1J* dl; G2J* dr; int32_t rc;
    if ((rc = upload<G1J>(e, e->jacG1, l, nl, &dl)) || (rc = upload<G2J>(e, e->jacG2, r, nr, &dr))) return rc;
    if ((rc = e->affG1.reserve(std::max<size_t>(nl, 1) * sizeof(G1A))) || (rc = e->affG2.reserve(std::max<size_t>(nl, 1) * sizeof(G2A)))) return rc;
    if ((rc = e->normalize_dev<Fp>(dl, nl, e->affG1.as<G1A>())) || (rc = e->normalize_dev<Fp2>(dr, nr, e->affG2.as<G2A>()))) return rc;
    Fp12 rows[N_LINES];
    const G1A* as[1] = {e->affG1.as<G1A>()}; const G2A* bs[1] = {e->affG2.as<G2A>()};
    if ((rc = e->step_products(as, bs, 1, nl, rows))) return rc;
    const Fp12 z = miller_combine(rows);
    std::memcpy(out, &z, sizeof z);
    e->collect_kernel_stats();
    return RIPP_OK;
}
API int32_t ripp_pairing_product_coeffs_a(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n, ripp_gt* out) {
    LOCK; ENGINE; if (!out || (n && (!a || !b || !r))) return RIPP_ERR_ARG;
    G1A* da; G2A* db; Fr* dr; int32_t rc;
    if ((rc = upload<G1A>(e, e->tmpA, a, n, &da))) return rc;
    if ((rc = upload<G2A>(e, e->tmpB, b, n, &db))) return rc;
    if ((rc = upload<Fr>(e, e->tmpR, r, n, &dr))) return rc;
    if ((rc = e->jacG1.reserve(std::max<size_t>(n, 1) * sizeof(G1J)))) return rc;
    if ((rc = e->affG1.reserve(std::max<size_t>(n, 1) * sizeof(G1A)))) return rc;
    if (n) {
        if ((rc = e->scale_g1_dev(da, 1u, dr, n, e->jacG1.as<G1J>()))) return rc;
        if ((rc = e->normalize_dev<Fp>(e->jacG1.as<G1J>(), n, e->affG1.as<G1A>()))) return rc;
    }
    return pairing_product_dev(e, e->affG1.as<G1A>(), db, n, out);
}


// ---- MSM -----------------------------------------------------------------------------------------------------------
extern "C++" {
template <class F, bool JAC> static int32_t msm_on(Engine* e, const void* bases, const ripp_fr* scalars, size_t nl, Jac<F>* out) {      // the MSM of nl terms on engine e (its device is current)
    Jac<F> res = jac_inf<F>();
    if (nl) {
        int32_t rc; Fr* ds;
        if ((rc = upload<Fr>(e, e->tmpR, scalars, nl, &ds))) return rc;
        DevBuf& jac = std::is_same<F, Fp>::value ? e->jacG1 : e->jacG2;
        DevBuf& aff = std::is_same<F, Fp>::value ? e->affG1 : e->affG2;
        if ((rc = aff.reserve(nl * sizeof(Affine<F>))) || (JAC && (rc = jac.reserve(nl * sizeof(Jac<F>))))) return rc;
        Affine<F>* db = aff.as<Affine<F>>();
        // the bases travel on stream2 while the engine stream sorts the digits of the scalars (msm_launch); projective inputs are normalised there too
        const std::function<int32_t()> bases_arrive = [&]() -> int32_t {
            if (JAC) {
                HIPCHK(hipMemcpyAsync(jac.p, bases, nl * sizeof(Jac<F>), hipMemcpyHostToDevice, e->stream2));
                int32_t r2 = e->normalize_dev<F>(jac.as<Jac<F>>(), nl, db, e->stream2); if (r2) return r2;      // inner_products/src/lib.rs:140
            } else HIPCHK(hipMemcpyAsync(db, bases, nl * sizeof(Affine<F>), hipMemcpyHostToDevice, e->stream2));
            HIPCHK(hipEventRecord(e->ev_join, e->stream2));
            HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join, 0));
            return RIPP_OK;
        };
        if (nl >= 2 && nl * sizeof(Affine<F>) >= e->msm_chunk_min * sizeof(G1A)) {          // (measured: pays from 2^20 G1 / 2^19 G2 bases, i.e. from ~96 MB of them)
            // Large host-slice MSMs in TWO halves on two streams (scratch sets 0 and 1, as the KZG openings use them): the second half of the bases crosses
            // PCIe while the first half's gathered additions run (the bases are 96 / 192 MB at n = 2^20: ~2 / ~4 ms of a 9.5 / 22 ms call in which only the
            // 0.35 ms digit sort was hidden), and the first half's latency-bound bucket stages run beside the second half's additions.  sum = MSM(lo) + MSM(hi).
            const size_t h = nl / 2;
            auto arrive = [&](size_t off, size_t cnt, hipStream_t st, hipEvent_t ev) -> int32_t {
                if (JAC) {
                    HIPCHK(hipMemcpyAsync(jac.as<Jac<F>>() + off, static_cast<const Jac<F>*>(bases) + off, cnt * sizeof(Jac<F>), hipMemcpyHostToDevice, e->stream2));
                    int32_t r2 = e->normalize_dev<F>(jac.as<Jac<F>>() + off, cnt, db + off, e->stream2); if (r2) return r2;
                } else HIPCHK(hipMemcpyAsync(db + off, static_cast<const Affine<F>*>(bases) + off, cnt * sizeof(Affine<F>), hipMemcpyHostToDevice, e->stream2));
                HIPCHK(hipEventRecord(ev, e->stream2));
                HIPCHK(hipStreamWaitEvent(st, ev, 0));
                return RIPP_OK;
            };
            const std::function<int32_t()> arrive_lo = [&]() { return arrive(0, h, e->stream, e->ev_join); };
            const std::function<int32_t()> arrive_hi = [&]() { return arrive(h, nl - h, e->stream3, e->ev_join3); };
            HIPCHK(hipEventRecord(e->ev_fork, e->stream));                                  // the scalars are on the device
            HIPCHK(hipStreamWaitEvent(e->stream3, e->ev_fork, 0));
            if ((rc = e->msm_launch<F>(e->msm_scratch[0], e->stream, db, ds, h, &arrive_lo))) return rc;
            if ((rc = e->msm_launch<F>(e->msm_scratch[1], e->stream3, db + h, ds + h, nl - h, &arrive_hi))) return rc;
            if ((rc = e->sync())) return rc;
            HIPCHK(hipStreamSynchronize(e->stream3)); HIPCHK(hipStreamSynchronize(e->stream2));
            res = add(*reinterpret_cast<const Jac<F>*>(e->msm_scratch[0].host_out), *reinterpret_cast<const Jac<F>*>(e->msm_scratch[1].host_out));
        }
        else if ((rc = e->msm_dev<F>(db, ds, nl, &res, &bases_arrive))) return rc;
    }
    *out = res;
    return RIPP_OK;
}
template <class F, bool JAC> static int32_t msm_impl(const void* bases, size_t nl, const ripp_fr* scalars, size_t nr, void* out) {
    if (nl != nr) { set_err("left length, right length: " + std::to_string(nl) + ", " + std::to_string(nr)); return RIPP_ERR_LENGTH; }
    LOCK; ENGINE; if (!out || (nl && (!bases || !scalars))) return RIPP_ERR_ARG;
    std::vector<DevSlot> sl; int32_t rc = device_slots(e, nl, MSM_TERMS_PER_DEVICE_MIN, &sl); if (rc) return rc;
    std::vector<Jac<F>> part(sl.size());
    using B = typename std::conditional<JAC, Jac<F>, Affine<F>>::type;
    if ((rc = run_on_devices(e, sl, [&](Engine* ee, size_t off, size_t cnt, int d) { return msm_on<F, JAC>(ee, static_cast<const B*>(bases) + off, scalars + off, cnt, &part[(size_t)d]); }))) return rc;
    Jac<F> res = part[0];
    for (size_t d = 1; d < sl.size(); ++d) res = add(res, part[d]);
    std::memcpy(out, &res, sizeof(Jac<F>));
    return RIPP_OK;
}
}  // extern "C++"
API int32_t ripp_msm_g1_j(const ripp_g1j* b, size_t nl, const ripp_fr* s, size_t nr, ripp_g1j* out) { return msm_impl<Fp, true>(b, nl, s, nr, out); }
API int32_t ripp_msm_g2_j(const ripp_g2j* b, size_t nl, const ripp_fr* s, size_t nr, ripp_g2j* out) { return msm_impl<Fp2, true>(b, nl, s, nr, out); }
API int32_t ripp_msm_g1_a(const ripp_g1a* b, const ripp_fr* s, size_t n, ripp_g1j* out) { return msm_impl<Fp, false>(b, n, s, n, out); }
API int32_t ripp_msm_g2_a(const ripp_g2a* b, const ripp_fr* s, size_t n, ripp_g2j* out) { return msm_impl<Fp2, false>(b, n, s, n, out); }

// ---- ScalarInnerProduct (inner_products/src/lib.rs:144-166) ----------------------------------------------------------------
API int32_t ripp_scalar_inner_product(const ripp_fr* l, size_t nl, const ripp_fr* r, size_t nr, ripp_fr* out) {
    if (nl != nr) { set_err("left length, right length: " + std::to_string(nl) + ", " + std::to_string(nr)); return RIPP_ERR_LENGTH; }
    LOCK; ENGINE; if (!out || (nl && (!l || !r))) return RIPP_ERR_ARG;
    Fr acc = Fr::zero();
    if (nl) {
        const unsigned blocks = std::min<unsigned>(1024, nblk(nl, 256));
        Fr *dl, *dr; int32_t rc;
        if ((rc = upload<Fr>(e, e->tmpR, l, nl, &dl)) || (rc = upload<Fr>(e, e->tmpA, r, nl, &dr)) || (rc = e->tmpB.reserve(blocks * sizeof(Fr)))) return rc;
        hipLaunchKernelGGL(k_fr_dot, dim3(blocks), dim3(256), 0, e->stream, dl, dr, (uint32_t)nl, e->tmpB.as<Fr>());
        HIPCHK(hipGetLastError());
        std::vector<Fr> part(blocks);
        HIPCHK(hipMemcpyAsync(part.data(), e->tmpB.p, blocks * sizeof(Fr), hipMemcpyDeviceToHost, e->stream));
        if ((rc = e->sync())) return rc;
        for (const Fr& p : part) acc = add(acc, p);
    }
    std::memcpy(out, &acc, sizeof acc);
    return RIPP_OK;
}

// ---- SIPP verifier (sipp/src/lib.rs:109-180) ----------------------------------------------------------------------------
API int32_t ripp_sipp_verify(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n, const ripp_gt* claimed, const ripp_gt* proof, size_t proof_rounds, int32_t* accept) {
    if (!a || !b || !r || !claimed || !proof || !accept) return RIPP_ERR_ARG;
    if (n < 2 || (n & (n - 1))) return RIPP_ERR_POW2;                     // asserts at :118-119
    size_t lg = 0; while (((size_t)1 << lg) < n) ++lg;
    if (proof_rounds != lg) return RIPP_ERR_ARG;                          // :122-123
    LOCK; ENGINE;
    const double tv0 = now_ms(); double tv1 = tv0;
    auto mark = [&](const char* what) { if (trace_on()) { const double t = now_ms(); fprintf(stderr, "[ripp] verify %-10s %.2f ms (t=%.2f)\n", what, t - tv1, t - tv0); tv1 = t; } };
    // the bases do not depend on the challenges: their upload runs while the host hashes the statement (the verifier's serial floor too)
    G1A* da; G2A* db; int32_t rc;
    if ((rc = upload<G1A>(e, e->affG1, a, n, &da)) || (rc = upload<G2A>(e, e->affG2, b, n, &db))) return rc;
    uint8_t digest[32];
    if ((rc = ripp_sipp_seed_digest(a, b, r, n, claimed, digest))) return rc;      // :126-132
    mark("digest");
    fs::FiatShamirRng rng; rng.from_digest(digest);
    const Fp12* pr = reinterpret_cast<const Fp12*>(proof);
    std::vector<Fp12> P(2 * lg); std::memcpy(P.data(), pr, 2 * lg * sizeof(Fp12));
    std::vector<Fr> xs(lg), xinv(lg);
    for (size_t j = 0; j < lg; ++j) { xs[j] = fs::sipp_challenge(rng, P[2 * j], P[2 * j + 1]); xinv[j] = inv(xs[j]); }   // :134-149
    // z' = z * prod z_l^x z_r^(x^-1)  (:151-158): 2 log n independent GT exponentiations on the host workers.  The proof's elements are
    // untrusted Fp12 values, so this uses the plain square-and-multiply (gt_pow_host's cyclotomic squarings assume GT membership).
    // They are queued BEFORE the scalar vectors are built when that build is single-threaded (it then overlaps them), and AFTER it when
    // its chunks go through the same FIFO pool (n >= 2^14: the chunks would wait 2-3 ms behind the powers, which overlap the MSMs instead).
    auto gt_pow = [](const Fp12& x, const Fr& k) { const Fr c = from_mont(k); Fp12 acc = Fp12::one(); bool st = false;
        for (int i = 255; i >= 0; --i) { if (st) acc = sqr(acc); if ((c.l[i >> 5] >> (i & 31)) & 1u) { acc = st ? mul(acc, x) : x; st = true; } } return acc; };
    std::vector<std::future<Fp12>> pw;
    // the tasks read P / xs / xinv by reference: every exit path (the HIP error returns included) waits for them first
    struct Drain { std::vector<std::future<Fp12>>& v; ~Drain() { for (auto& f : v) if (f.valid()) f.wait(); } } drain{pw};
    auto submit_powers = [&]() {
        for (size_t j = 0; j < lg; ++j) {
            pw.push_back(host_pool().submit([&, j]() { return gt_pow(P[2 * j], xs[j]); }));
            pw.push_back(host_pool().submit([&, j]() { return gt_pow(P[2 * j + 1], xinv[j]); }));
        }
    };
    constexpr size_t POOLED_MIN = (size_t)1 << 14;
    if (n < POOLED_MIN) submit_powers();
    // s_i = r_i * prod_{j : bit (lg-1-j) of i set} x_j ;  s_inv likewise (:160-172), built by doubling; long levels are split over the workers
    if ((rc = e->stage[0].reserve(n * sizeof(Fr))) || (rc = e->stage[1].reserve(n * sizeof(Fr)))) return rc;
    Fr* const s = e->stage[0].as<Fr>(); Fr* const si = e->stage[1].as<Fr>(); s[0] = Fr::one(); si[0] = Fr::one();
    const Fr* rr = reinterpret_cast<const Fr*>(r);
    auto parallel_for = [](size_t count, const std::function<void(size_t, size_t)>& body) {
        const size_t chunks = count >= POOLED_MIN ? 6 : 1, per = (count + chunks - 1) / chunks;
        std::vector<std::future<void>> f;
        for (size_t c = 1; c < chunks; ++c) { const size_t lo = c * per, hi = std::min(count, lo + per); if (lo < hi) f.push_back(host_pool().submit([&body, lo, hi]() { body(lo, hi); })); }
        body(0, std::min(count, per));
        for (auto& x : f) x.get();
    };
    for (size_t j = lg; j-- > 0;) {
        const size_t bit = (size_t)1 << (lg - 1 - j); const Fr xj = xs[j], xij = xinv[j];
        parallel_for(bit, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) { s[i + bit] = mul(s[i], xj); si[i + bit] = mul(si[i], xij); } });
    }
    const bool aligned = ((uintptr_t)r & 15u) == 0;
    std::vector<Fr> rcopy; if (!aligned) { rcopy.resize(n); std::memcpy(rcopy.data(), r, n * sizeof(Fr)); rr = rcopy.data(); }
    parallel_for(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) s[i] = mul(s[i], rr[i]); });
    if (n >= POOLED_MIN) submit_powers();
    mark("scalars");
    // the two MSMs (:174-175) side by side on two streams against the bases already resident
    if ((rc = e->kzg_q[0].reserve(n * sizeof(Fr))) || (rc = e->kzg_q[1].reserve(n * sizeof(Fr)))) return rc;
    if ((rc = e->stage[0].send(e->kzg_q[0].p, n * sizeof(Fr), e->stream)) || (rc = e->stage[1].send(e->kzg_q[1].p, n * sizeof(Fr), e->stream))) return rc;
    if ((rc = e->sync())) return rc;
    mark("h2d");
    if ((rc = e->msm_launch<Fp>(e->msm_scratch[1], e->stream3, da, e->kzg_q[0].as<Fr>(), n))) return rc;
    if ((rc = e->msm_launch<Fp2>(e->msm_scratch[0], e->stream, db, e->kzg_q[1].as<Fr>(), n))) return rc;
    mark("msm launch");
    if ((rc = e->sync())) return rc; HIPCHK(hipStreamSynchronize(e->stream3));
    mark("msm sync");
    const G1A apa = to_affine(*reinterpret_cast<const G1J*>(e->msm_scratch[1].host_out)); const G2A bpa = to_affine(*reinterpret_cast<const G2J*>(e->msm_scratch[0].host_out));
    Fp12 zp; std::memcpy(&zp, claimed, sizeof zp);
    for (auto& f : pw) zp = mul(zp, f.get());
    mark("gt powers");
    G1A* d1; G2A* d2; ripp_gt e12;
    if ((rc = upload<G1A>(e, e->tmpA, &apa, 1, &d1)) || (rc = upload<G2A>(e, e->tmpB, &bpa, 1, &d2))) return rc;
    if ((rc = pairing_product_dev(e, d1, d2, 1, &e12))) return rc;                                                       // :177
    mark("pairing");
    Fp12 ev; std::memcpy(&ev, &e12, sizeof ev);
    *accept = (ev == zp) ? 1 : 0;
    return RIPP_OK;
}


#include "comm_core.inc"     // RCCL / callback communicator (ripp_comm_*), used by the sharded provers below
#include "tipa_api.inc"      // GIPA / TIPA / TIPAWithSSM provers, aggregate_proofs, verifiers

// ---- SIPP ----------------------------------------------------------------------------------------------------------
// borrow_value != nullptr: one-shot proof -- the caller's (16-byte aligned) buffers outlive the job, so the statement hash runs on them
// in place and starts BEFORE the upload (it is the critical path: ~0.3 s at n = 2^20) instead of after a 336 MB host copy
// (full_a / full_b / full_r, n_full): the statement the hash runs over -- the shard itself for world == 1, the FULL statement on rank 0 of a sharded one-shot proof
static void job_buffers(ripp_sipp_job* j, DevBuf* (&d)[15], PinBuf* (&p)[4]) {
    DevBuf* dl[15] = {&j->a0, &j->b0, &j->r0, &j->a, &j->b, &j->a_next, &j->b_next, &j->jac1, &j->jac2, &j->a_pow, &j->b_pow, &j->a_pow_h, &j->b_pow_h, &j->parts1, &j->parts2};
    PinBuf* pl[4] = {&j->tp_rows[0], &j->tp_rows[1], &j->look_rows[0], &j->look_rows[1]};
    for (int i = 0; i < 15; ++i) d[i] = dl[i];
    for (int i = 0; i < 4; ++i) p[i] = pl[i];
}
static void job_release_buffers(ripp_sipp_job* j) { DevBuf* d[15]; PinBuf* p[4]; job_buffers(j, d, p); for (DevBuf* b : d) b->release(); for (PinBuf* b : p) b->release(); }
// (on engine e, whose device is current and whose caller holds the library's lock: the C-ABI wrapper below, or one in-process rank of sipp_prove_devices)
static int32_t sipp_job_create_on(Engine* e, const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n_local, int32_t rank, int32_t world, const ripp_gt* borrow_value, ripp_sipp_job** job,
                                  const ripp_g1a* full_a = nullptr, const ripp_g2a* full_b = nullptr, const ripp_fr* full_r = nullptr, size_t n_full = 0, bool one_shot = false) {
    if (!a || !b || !r || !job || n_local == 0 || world < 1 || rank < 0 || rank >= world) return RIPP_ERR_ARG;
    if (n_local & (n_local - 1)) return RIPP_ERR_POW2;
    ripp_sipp_job* j = new ripp_sipp_job();
    if (one_shot && e->job_cache.full) {                      // the buffers the previous one-shot proof left behind (sipp_job_retire)
        DevBuf* d[15]; PinBuf* p[4]; job_buffers(j, d, p);
        for (int i = 0; i < 15; ++i) std::swap(*d[i], e->job_cache.d[i]);
        for (int i = 0; i < 4; ++i) std::swap(*p[i], e->job_cache.p[i]);
        e->job_cache.full = false;
    }
    struct Live { bool keep = false; Live() { ++g_live_handles; } ~Live() { if (!keep) --g_live_handles; } } live;
    j->n_local = n_local; j->rank = rank; j->world = world; j->world0 = world;
    if (world == 1 && !full_a) { full_a = a; full_b = b; full_r = r; n_full = n_local; }
    const bool borrow = borrow_value && full_a && full_b && full_r && n_full == n_local * (size_t)world && rank == 0 && (((uintptr_t)full_a | (uintptr_t)full_b | (uintptr_t)full_r) & 15u) == 0;
    if (borrow) {
        j->ha_ext = reinterpret_cast<const G1A*>(full_a); j->hb_ext = reinterpret_cast<const G2A*>(full_b); j->hr_ext = reinterpret_cast<const Fr*>(full_r); j->hash_n = n_full;
        Fp12 v; std::memcpy(&v, borrow_value, sizeof v);
        job_start_hash(j, v); j->hash_prestarted = true;
    }
    int32_t rc;
    if ((rc = j->a0.reserve(n_local * sizeof(G1A))) || (rc = j->b0.reserve(n_local * sizeof(G2A))) || (rc = j->r0.reserve(n_local * sizeof(Fr)))) { if (j->hash_thread.joinable()) j->hash_thread.join(); job_release_buffers(j); delete j; return rc; }
    {   // on failure: join the hash thread (it reads the CALLER's buffers in borrow mode) and free the job before returning
        hipError_t he = hipMemcpyAsync(j->a0.p, a, n_local * sizeof(G1A), hipMemcpyHostToDevice, e->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(j->b0.p, b, n_local * sizeof(G2A), hipMemcpyHostToDevice, e->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(j->r0.p, r, n_local * sizeof(Fr), hipMemcpyHostToDevice, e->stream);
        if (he != hipSuccess) {
            set_err(std::string("statement upload: ") + hipGetErrorString(he));
            if (j->hash_thread.joinable()) j->hash_thread.join();
            job_release_buffers(j);
            delete j; return RIPP_ERR_DEVICE;
        }
    }
    if (world == 1 && !borrow) {   // single-GPU jobs hash their own statement; keep the host image
        j->ha.resize(n_local); j->hb.resize(n_local); j->hr.resize(n_local);
        std::memcpy(j->ha.data(), a, n_local * sizeof(G1A)); std::memcpy(j->hb.data(), b, n_local * sizeof(G2A)); std::memcpy(j->hr.data(), r, n_local * sizeof(Fr));
    }
    if ((rc = e->sync())) { if (j->hash_thread.joinable()) j->hash_thread.join(); job_release_buffers(j); delete j; return rc; }
    live.keep = true;
    *job = j; return RIPP_OK;
}
static int32_t sipp_job_create_impl(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n_local, int32_t rank, int32_t world, const ripp_gt* borrow_value, ripp_sipp_job** job,
                                    const ripp_g1a* full_a = nullptr, const ripp_g2a* full_b = nullptr, const ripp_fr* full_r = nullptr, size_t n_full = 0, bool one_shot = false) {
    LOCK; ENGINE;
    return sipp_job_create_on(e, a, b, r, n_local, rank, world, borrow_value, job, full_a, full_b, full_r, n_full, one_shot);
}
API int32_t ripp_sipp_job_create(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n_local, int32_t rank, int32_t world, ripp_sipp_job** job) {
    return sipp_job_create_impl(a, b, r, n_local, rank, world, nullptr, job);
}
API void ripp_sipp_job_destroy(ripp_sipp_job* j) {
    if (!j) return; LOCK;
    if (j->hash_thread.joinable()) j->hash_thread.join();
    job_release_buffers(j);
    if (g_engine && g_engine->tab_owner == j) g_engine->tab_owner = nullptr;
    delete j; --g_live_handles;
}
// the end of a one-shot proof: the job goes, its buffers stay with the engine for the next one-shot call (Engine::job_cache)
// the same for a job of engine e whose caller holds the lock (an in-process rank): park the buffers with THAT engine, then free the job
static void sipp_job_retire_on(Engine* e, ripp_sipp_job* j) {
    if (!j) return;
    if (j->hash_thread.joinable()) j->hash_thread.join();
    if (!e->job_cache.full && !e->sw.no_job_cache) {
        DevBuf* d[15]; PinBuf* p[4]; job_buffers(j, d, p);
        bool idle = true; for (PinBuf* b : p) if (b->wait() != RIPP_OK) idle = false;
        if (idle) {
            for (int i = 0; i < 15; ++i) std::swap(*d[i], e->job_cache.d[i]);
            for (int i = 0; i < 4; ++i) std::swap(*p[i], e->job_cache.p[i]);
            e->job_cache.full = true;
        }
    }
    job_release_buffers(j);
    if (e->tab_owner == j) e->tab_owner = nullptr;
    delete j; --g_live_handles;
}
static void sipp_job_retire(ripp_sipp_job* j) {
    if (!j) return;
    {   LOCK;
        if (j->hash_thread.joinable()) j->hash_thread.join();
        Engine* e = g_engine;
        if (e && !e->job_cache.full && !e->sw.no_job_cache) {
            DevBuf* d[15]; PinBuf* p[4]; job_buffers(j, d, p);
            bool idle = true; for (PinBuf* b : p) if (b->wait() != RIPP_OK) idle = false;      // (nothing is in flight after a proof: sipp_prove_core's Quiesce)
            if (idle) {
                for (int i = 0; i < 15; ++i) std::swap(*d[i], e->job_cache.d[i]);
                for (int i = 0; i < 4; ++i) std::swap(*p[i], e->job_cache.p[i]);
                e->job_cache.full = true;
            }
        }
    }
    ripp_sipp_job_destroy(j);
}
API int32_t ripp_sipp_job_begin(ripp_sipp_job* j) { LOCK; ENGINE; if (!j) return RIPP_ERR_ARG; return job_begin(e, j); }
API size_t ripp_sipp_job_rounds_left(const ripp_sipp_job* j) { if (!j) return 0; size_t total = j->len * (size_t)j->world, r = 0; while (total > 1) { total >>= 1; ++r; } return r; }
API int32_t ripp_sipp_job_round_partials(ripp_sipp_job* j, ripp_gt* partials) {
    LOCK; ENGINE; if (!j || !partials) return RIPP_ERR_ARG;
    if (j->len < 2) { set_err("shard exhausted: gather the remaining elements onto one rank"); return RIPP_ERR_ARG; }
    Fp12 rows[2 * N_LINES];
    int32_t rc = job_round_partials(e, j, rows); if (rc) return rc;
    // round 0 of a staged (sharded) proof: the challenge needs the digest of the WHOLE statement, which rank 0 is still hashing -- build this
    // shard's fold tables behind the products, exactly as ripp_sipp_job_prove does on one GPU
    if (!j->seeded && j->len == j->n_local && (rc = job_precompute_round0(e, j))) return rc;
    if ((rc = job_precompute_vm(e, j))) return rc;          // small rounds: second fold bases on the VM while the ranks exchange and finish the values
    auto fut = host_pool().submit([&rows]() { return miller_combine(rows + N_LINES); });
    const Fp12 ml = miller_combine(rows), mr = fut.get();
    std::memcpy(&partials[0], &ml, sizeof ml); std::memcpy(&partials[1], &mr, sizeof mr);
    return RIPP_OK;
}
API int32_t ripp_sipp_job_round_finish(ripp_sipp_job* j, const ripp_gt* combined, const uint8_t seed_digest[32], ripp_gt* z_l, ripp_gt* z_r, ripp_fr* x) {
    LOCK; ENGINE; if (!j || !combined || !z_l || !z_r || !x) return RIPP_ERR_ARG;
    const double t0 = now_ms();
    Fp12 mv[2]; std::memcpy(mv, combined, sizeof mv);
    auto fut = host_pool().submit([&mv]() { return final_exponentiation(mv[1]); });
    const Fp12 zl = final_exponentiation(mv[0]);
    const Fp12 zr = fut.get();
    if (!j->seeded) { if (!seed_digest) return RIPP_ERR_ARG; j->rng.from_digest(seed_digest); j->seeded = true; }
    const Fr xc = fs::sipp_challenge(j->rng, zl, zr);
    e->stats.host_ms += now_ms() - t0;
    std::memcpy(z_l, &zl, sizeof(Fp12)); std::memcpy(z_r, &zr, sizeof(Fp12)); std::memcpy(x, &xc, sizeof(Fr));
    return job_fold(e, j, xc);
}
API size_t ripp_sipp_job_local_len(const ripp_sipp_job* j) { return j ? j->len : 0; }
API int32_t ripp_sipp_job_export(ripp_sipp_job* j, ripp_g1a* a_out, ripp_g2a* b_out) {
    LOCK; ENGINE; if (!j || !a_out || !b_out) return RIPP_ERR_ARG;
    if (j->len == 0) { set_err("ripp_sipp_job_export: the job's working vectors were consumed by a whole proof (ripp_sipp_job_begin re-arms it)"); return RIPP_ERR_ARG; }
    HIPCHK(hipMemcpyAsync(a_out, j->a.p, j->len * sizeof(G1A), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(b_out, j->b.p, j->len * sizeof(G2A), hipMemcpyDeviceToHost, e->stream));
    return e->sync();
}
API int32_t ripp_sipp_job_import(ripp_sipp_job* j, const ripp_g1a* a, const ripp_g2a* b, size_t len) {
    LOCK; ENGINE; if (!j || !a || !b || len == 0 || (len & (len - 1))) return RIPP_ERR_ARG;
    int32_t rc;
    if ((rc = j->a.reserve(len * sizeof(G1A))) || (rc = j->b.reserve(len * sizeof(G2A))) || (rc = j->a_next.reserve(len * sizeof(G1A))) ||
        (rc = j->b_next.reserve(len * sizeof(G2A))) || (rc = j->jac1.reserve(len * sizeof(G1J))) || (rc = j->jac2.reserve(len * sizeof(G2J)))) return rc;
    HIPCHK(hipMemcpyAsync(j->a.p, a, len * sizeof(G1A), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(j->b.p, b, len * sizeof(G2A), hipMemcpyHostToDevice, e->stream));
    if ((rc = e->sync())) return rc;
    j->len = len; j->world = 1;
    return RIPP_OK;
}
API int32_t ripp_combine_partials(const ripp_gt* gathered, int32_t world, size_t count, ripp_gt* out) {
    if (!gathered || !out || world < 1) return RIPP_ERR_ARG;
    const Fp12* g = reinterpret_cast<const Fp12*>(gathered);
    for (size_t k = 0; k < count; ++k) {
        Fp12 acc; std::memcpy(&acc, &g[k], sizeof acc);
        for (int w = 1; w < world; ++w) { Fp12 t; std::memcpy(&t, &g[(size_t)w * count + k], sizeof t); acc = mul(acc, t); }
        std::memcpy(&out[k], &acc, sizeof acc);
    }
    return RIPP_OK;
}
API int32_t ripp_sipp_job_stats(const ripp_sipp_job* j, ripp_stats* st) { LOCK; ENGINE; if (!j || !st) return RIPP_ERR_ARG; e->collect_kernel_stats(); *st = e->stats; return RIPP_OK; }

// ---- pipelined tail rounds -------------------------------------------------------------------------------------------------------------
// A latency-bound round is  products (0.5 ms, device) -> final exponentiations + hash (0.65 ms, host) -> fold (0.7 ms, device): 1.8 ms in which
// device and host wait for each other.  With the quarters a = [a0|a1|a2|a3], b = [b0|b1|b2|b3] of the CURRENT vectors and this round's
// challenge x, the NEXT round's values are (bilinearity, as in job_preevaluate_round1)
//     z_l' = E(a1,b0) E(a1,b2)^(1/x) E(a3,b0)^x E(a3,b2),      z_r' = E(a0,b1) E(a0,b3)^(1/x) E(a2,b1)^x E(a2,b3),
// eight quarter-size products that need no challenge: four times the pairs, which costs nothing while a round does not fill the chip.  They
// are enqueued BEHIND the previous fold; the fold itself is no longer waited for: once x is known the host goes straight to the next
// round's six final exponentiations and four GT powers while the device folds and evaluates the round after that.  Same group elements,
// same proof bytes.
static bool tail_pipe_ok(const Engine* e, const ripp_sipp_job* j) {
    return j->len >= 4 && j->len <= e->tail_pipe_max && j->len / 2 <= e->vm_fold_max && 2 * j->len <= e->max_pairs_per_batch && !j->bs_on && j->xs_enabled && !e->sw.no_vm && !e->sw.no_precompute && !e->sw.no_endo;
}
// enqueue the eight quarter products of the current vectors: they become round `for_round`'s values
static int32_t job_tail_enqueue(Engine* e, ripp_sipp_job* j, size_t for_round) {
    const size_t q = j->len / 4; const int slot = (int)(for_round & 1);
    const G1A* a = j->a.as<G1A>(); const G2A* b = j->b.as<G2A>();
    const G1A* as[8] = {a + q, a + q, a + 3 * q, a + 3 * q, a, a, a + 2 * q, a + 2 * q};
    const G2A* bs[8] = {b, b + 2 * q, b, b + 2 * q, b + q, b + 3 * q, b + q, b + 3 * q};
    PinBuf& buf = j->tp_rows[slot];
    int32_t rc = buf.reserve(8 * N_LINES * sizeof(Fp12)); if (rc) return rc;
    if ((rc = e->enqueue_products(as, bs, 8, 0, q, buf.as<Fp12>()))) return rc;
    HIPCHK(hipEventRecord(buf.ev, e->stream)); buf.pending = true;
    j->tp_round[slot] = for_round;
    return RIPP_OK;
}
// (z_l, z_r) of round `round` from its enqueued quarter products and the previous challenge
static int32_t job_tail_values(ripp_sipp_job* j, size_t round, const Fr& x_prev, Fp12* zl, Fp12* zr) {
    const int slot = (int)(round & 1);
    PinBuf& buf = j->tp_rows[slot];
    const double tw0 = now_ms();
    int32_t rc = buf.wait(); if (rc) return rc;
    const double tw1 = now_ms();
    j->tp_round[slot] = ~(size_t)0;
    const Fp12* rows = buf.as<Fp12>();
    Fp12 T[6];          // per side: E0 * E3 (one final exponentiation for the pair), E1, E2
    // each value in `parts` bit ranges of the Miller recurrence, every range with its own final exponentiation on its own worker, joined by cyclotomic
    // squarings (pairing_values has the identity): 2 ranges where the pool has the workers for 12 tasks
    const int parts = host_pool().size() >= 11 ? 2 : 1;
    Fp12 E[6][2];
    auto range_of = [parts](int g, int* hi, int* lo) { *hi = 62 - (63 * g) / parts; *lo = 62 - (63 * (g + 1)) / parts + 1; };
    host_pool().parallel(6 * parts, [&](int u) {
        const int t = u / parts, g = u % parts;
        int hi, lo; range_of(g, &hi, &lo);
        const Fp12* base = rows + (size_t)(t / 3) * 4 * N_LINES; const int kind = t % 3;
        if (kind == 0) {
            int s0 = 0; for (int b = 62; b > hi; --b) s0 += 1 + (int)((BLS_X_ABS >> b) & 1);      // the rows this range consumes: [s0, s1)
            int s1 = s0; for (int b = hi; b >= lo; --b) s1 += 1 + (int)((BLS_X_ABS >> b) & 1);
            Fp12 prod[N_LINES];
            for (int s2 = s0; s2 < s1; ++s2) prod[s2] = mul(base[s2], base[3 * N_LINES + s2]);
            E[t][g] = final_exponentiation(miller_combine_range(prod, hi, lo));
        } else E[t][g] = final_exponentiation(miller_combine_range(base + (size_t)kind * N_LINES, hi, lo));
    });
    host_pool().parallel(6, [&](int t) {
        Fp12 c = E[t][0];
        for (int g = 1; g < parts; ++g) { int hi, lo; range_of(g, &hi, &lo); for (int b = hi; b >= lo; --b) c = cyclotomic_sqr(c); c = mul(c, E[t][g]); }
        T[t] = BLS_X_NEG ? conj(c) : c;
    });
    const double tw2 = now_ms();
    // per side: E1^(1/x) as two tasks of two digit strings each (1/x is full width: four strings), E2^x as one (x is 128 bits: two strings)
    const GlsDigits gx = gls_digits(x_prev), gxi = gls_digits(inv(x_prev));
    if (parts == 1) {
        Fp12 P[6];
        host_pool().parallel(6, [&](int t) {
            const int side = t / 3, part = t % 3;
            P[t] = part == 2 ? gt_pow_gls_strings(T[3 * side + 2], gx, 15u) : gt_pow_gls_strings(T[3 * side + 1], gxi, part == 0 ? 3u : 12u);
        });
        *zl = mul(mul(T[0], P[0]), mul(P[1], P[2])); *zr = mul(mul(T[3], P[3]), mul(P[4], P[5]));
    } else {            // twelve workers: one digit string of 1/x per task, two tasks for x (whichever of its four strings are in use)
        Fp12 P[12];
        static const unsigned mask[6] = {1u, 2u, 4u, 8u, 5u, 10u};
        host_pool().parallel(12, [&](int t) {
            const int side = t / 6, part = t % 6;
            P[t] = part >= 4 ? gt_pow_gls_strings(T[3 * side + 2], gx, mask[part]) : gt_pow_gls_strings(T[3 * side + 1], gxi, mask[part]);
        });
        *zl = mul(mul(mul(T[0], P[0]), mul(P[1], P[2])), mul(P[3], mul(P[4], P[5])));
        *zr = mul(mul(mul(T[3], P[6]), mul(P[7], P[8])), mul(P[9], mul(P[10], P[11])));
    }
    if (trace_on()) fprintf(stderr, "[ripp] tail values: waited %.2f ms for the device, final exponentiations %.2f ms, powers %.2f ms\n", tw1 - tw0, tw2 - tw1, now_ms() - tw2);
    return RIPP_OK;
}

// ---- look-ahead: the values of rounds 1..k from the ROUND-0 vectors, evaluated in the hash window ----------------------------------------
// With the current vectors cut into 2^(R+1) blocks A_0.., B_0.. (block index = the top R+1 bits of the element index), R folds with the
// challenges x_0 .. x_(R-1) give   a^(R)_c = sum_e (prod_t x_t^(e_t)) A_(e,c),   b^(R)_c = sum_f (prod_t x_t^(-f_t)) B_(f,c)   (e, f in {0,1}^R,
// (e,c) = the block index with bits e_0 .. e_(R-1), c), hence by bilinearity
//     z_l^(R) = E(a^(R)_1, b^(R)_0) = prod_(e,f) E(A_(e,1), B_(f,0))^(prod_t x_t^(e_t - f_t)),      z_r^(R) likewise with (c_a, c_b) = (0, 1):
// 4^R products of len / 2^(R+1) pairs each that need NO challenge.  Products with the same exponent vector d = e - f (3^R of them) are multiplied
// per step on the host before their ONE final exponentiation.  The statement hash (sequential Blake2s, ~0.32 s at n = 2^20) is the window:
// one GPU fits z_l of round 1 behind round 0 and the fold tables (the former job_preevaluate_round1); G ranks have a G times longer window
// relative to their shard and take both values of rounds 1..k (look_plan).  When x_t arrives, level t of every item is applied,
//     Z'[rest] = Z[d_t = -1, rest]^(1/x_t) * Z[0, rest] * Z[+1, rest]^(x_t),
// on the host workers (GT powers in base |x|: the values are outputs of our own final exponentiation): the item of round t + 1 first -- two
// powers per value on the critical path -- the deeper ones in the background.  Same group elements, hence the same proof bytes.
// In a sharded proof every rank does this on its shard: all maps involved are homomorphisms, the ranks' values multiply to the whole one.
constexpr int LOOK_MAX_R = 3;
static int pow3(int r) { int v = 1; while (r-- > 0) v *= 3; return v; }
static HostPool& look_pool() { static HostPool pool(3); return pool; }      // own workers: the statement hash's serialisation tasks must never queue behind these
// how many (round, side) items -- in the order (1,l) (1,r) (2,l) (2,r) .. -- fit the hash window.  Decided by rank 0 (it knows whether anybody hashes) and sent to the others.
// Returned in EIGHTHS of an item: 8 k + f = the first k items in full and f/8 of the pairs of the next one (the window is a fixed budget; a
// whole item of round R costs 2^(R-1) n pairs, so the last one is cut to what is left).
static int look_plan(const Engine* e, size_t n_local, int world, bool window) {
    if (!window || e->sw.no_precompute || e->sw.no_endo) return 0;
    if (e->look_eighths >= 0) return std::min(16 * LOOK_MAX_R, e->look_eighths);      // forced plan (ripp_config.look_eighths / RIPP_LOOK_EIGHTHS / RIPP_LOOK_ITEMS)
    const double share = e->ranks_per_device;                       // ranks sharing this rank's GPU (test rigs: several ranks on one device): their work adds up in the same window
    const double nl = (double)n_local * share, n = (double)n_local * world;
    if (n < (double)((size_t)1 << 17)) return 0;                    // small statements: the hash is done long before the GPU is
    // Rates: compiled-in for the FIRST proof of a process (2^20 pairs through lines + products ~80 ms with the carry-free kernels; the statement hash
    // 286 ms at n = 2^20 = 1.17 GB/s of Blake2s in situ with the x86-64 bulk loop, build round 5; 313-317 ms = 1.06-1.12 GB/s before), MEASURED afterwards: sipp_prove_core records this box's hash rate and this device's
    // pairing rate at the end of every large proof that hashed (Engine::cal_*), so the static plan the non-hashing ranks follow is priced with what
    // rank 0's box and GPU really do (boxes differ by +-3 % / +-5 %).  Scaling and the fold tables move with the GPU factor.
    const double ms_per_pair = (e->cal_ms_per_pair > 0 ? e->cal_ms_per_pair : 6.7e-5) * (1.0 + 0.01 * e->plan_derate_pct);      // (shared G2 chains: 138 ms for the 2^21 pair evaluations of round 0 + (1,l) at n = 2^20; plan_derate_pct: ripp_config, the plan of a slower device)
    const double gpu_f = ms_per_pair / 6.7e-5;
    const double hash_ms = n * 336.0 / (e->cal_hash_bytes_per_ms > 0 ? e->cal_hash_bytes_per_ms : 1.17e6);
    double budget = hash_ms - (nl * (3.2e-5 * gpu_f + ms_per_pair + 5.3e-5 * gpu_f) + 1.0);      // scaling, round 0, fold tables (measured at n = 2^20: 33 + 80 + 55 ms)
    int items = 0;
    for (int it = 0; it < 2 * LOOK_MAX_R; ++it) {
        const int R = it / 2 + 1;
        if (n_local >> (R + 1) < 1024) break;
        const double cost = nl * (double)(1 << (R - 1)) * ms_per_pair + 0.3 * pow3(R) * share;
        if (cost > budget) {                                        // an overrun costs its length 1 : 1, a pre-evaluated pair saves a quarter of its cost: cut, do not stretch
            const int f = (int)(8.0 * budget / cost + 0.5);
            return 8 * items + (f >= 2 ? std::min(f, 7) : 0);
        }
        budget -= cost; ++items;
    }
    return 8 * items;
}
// ms_per_pair: what round 0's products cost on THIS device a moment ago.  A rank that hashes the statement itself (rank 0, unless the plan is forced)
// sizes every item ADAPTIVELY: the hash thread counts the bytes it has consumed, so the time the window still has is known to a few per cent,
// the device queue is drained before an item is sized, and the item takes the whole / the fraction of its pairs that still fits -- boxes differ
// by +-3 % in hash speed and +-5 % in GPU speed, more than the static plan's margin.  Other ranks follow the plan rank 0 sent.
// first_item = 1: item (1,l) was already produced together with round 0 (job_round0_shared).
static int32_t job_lookahead(Engine* e, ripp_sipp_job* j, int eighths, bool forced, double ms_per_pair, int first_item = 0) {
    if (first_item == 0) j->look.clear();
    const bool adaptive = !forced && j->hash_total > 0 && !j->no_window && eighths > 0 && !e->look_static;
    // ranks that do not hash the statement themselves: rank 0 told them when it expects the digest (SippPlanMsg::hash_left_ms, its measured hash rate); they cut or
    // extend the plan it sent by THEIR clock and THEIR measured pairing rate -- a slower or a time-sliced device no longer overruns the window, a faster one uses it
    const bool by_deadline = !forced && !adaptive && j->look_deadline > 0 && !j->no_window && eighths > 0 && !e->look_static;
    const int items = (adaptive || by_deadline) ? std::min(2 * LOOK_MAX_R, eighths / 8 + 2) : (eighths + 7) / 8;      // adaptive: at most one whole item beyond what the static model expects
    if (items <= 0 || (j->digest_ready.load() && !forced)) return RIPP_OK;      // the hash is already done: nothing to hide the work behind (forced: RIPP_LOOK_ITEMS, tests)
    const double t0 = now_ms();
    const size_t len = j->len;
    const G1A* a = j->a.as<G1A>(); const G2A* b = j->b.as<G2A>();
    int32_t rc;
    for (int it = first_item; it < items; ++it) {
        const int R = it / 2 + 1, side = it & 1;
        const size_t qblk = len >> (R + 1);
        if (qblk == 0 || qblk > e->pairs_cap(std::min(e->max_pairs_per_batch, qblk * (size_t)MAX_PRODUCTS))) break;
        int frac = std::min(8, eighths - 8 * it);                        // static plan: the last item may be partial (the first frac/8 of every block's pairs)
        if (adaptive || by_deadline) {
            if (qblk < 1024) break;
            if ((rc = e->sync())) return rc;                             // the fold tables / the previous item have left the device: what follows starts now
            const uint64_t done = adaptive ? j->hash_done.load(std::memory_order_relaxed) : 1;
            if (adaptive && (j->digest_ready.load() || done == 0)) break;
            const double elapsed = adaptive ? now_ms() - j->hash_t0 : 0;
            double room = adaptive ? elapsed * (double)(j->hash_total - std::min(done, j->hash_total)) / (double)done - 3.0      // ms the hash still needs, minus the item's host work
                                   : j->look_deadline - now_ms() - 3.0;                                                            // ... by rank 0's estimate
            // the extrapolation is only trusted inside what a sequential Blake2s can plausibly need in total (0.9 - 1.25 GB/s): a progress counter that
            // lags (the hash thread descheduled, a burst of serialisation waits) must not make the window look longer than it can be
            if (adaptive) room = std::min(room, (double)j->hash_total / 0.9e6 - elapsed);
            // Cost of the item over q pairs per block: 4^R products of q pairs at the measured pairing rate -- or, when its launches leave stage 1
            // under-occupied (few chains, each walked for 2^R line sets: a lone wave needs ~3.5 ms + 0.3 ms per line set, up to MAX_SHARE of them, per launch whatever q is), that floor per
            // launch plus the rest of the pipeline (58 % of a pair evaluation).  Measured with recorded peers (profiles/r05_rank0_of_{2,8}_timeline.txt): item
            // (3,l) on 8 192-pair blocks took 59 ms where the rate alone says 37; 11/32 of it on 32 768-pair blocks 62 ms instead of 46 -- the overrun of
            // the two-rank window in build round 4's plan.
            const double mpp = std::max(ms_per_pair, 5.0e-5) * (1.0 + 0.01 * e->plan_derate_pct), nprod = (double)((size_t)1 << (2 * R));
            auto item_cost = [&](size_t qq) {
                const size_t per_launch = std::min<size_t>(MAX_PRODUCTS, std::max<size_t>(1, e->max_pairs_per_batch / std::max<size_t>(qq, 1)));
                const double launches = std::ceil(nprod / (double)per_launch), var = nprod * (double)qq * mpp;
                return std::max(var, launches * (3.5 + 0.3 * (double)std::min(1 << R, (int)MAX_SHARE)) + 0.58 * var);
            };
            const double cost = item_cost(qblk);
            if (trace_on() && adaptive) fprintf(stderr, "[ripp] look-ahead item (%d,%c): hash %.0f %% after %.1f ms, room %.1f ms, item %.1f ms\n", R, side ? 'r' : 'l', 100.0 * (double)done / (double)j->hash_total, elapsed, room, cost);
            if (trace_on() && by_deadline) fprintf(stderr, "[ripp] rank %d look-ahead item (%d,%c): %.1f ms to rank 0's expected digest, item %.1f ms\n", j->rank, R, side ? 'r' : 'l', room, cost);
            if (room <= 0) break;                                        // the window is over (or the clamp above says it must be): nothing more fits
            // item (1,r) is completed when at least 80 % of it fits: the missing part costs 4 x as much here as after the fold, but BOTH values of round 1
            // known at the digest is what lets rounds 0 and 1 fold in one pass (-8.5 ms): an overrun of up to ~14 ms pays
            if (room >= cost || (it == 1 && room >= 0.8 * cost)) frac = 8;
            else {                                                       // adaptive: the 32nds of the blocks that still fit (negative = in 32nds)
                int f32 = (int)(32.0 * room / cost);
                while (f32 > 0 && item_cost((qblk * (size_t)f32 / 32) & ~(size_t)63) > room) --f32;
                frac = -f32;
            }
            if (frac <= 0 && frac > -6) break;
        }
        const size_t q = frac >= 8 ? qblk : frac < 0 ? (qblk * (size_t)(-frac) / 32) & ~(size_t)63 : (qblk * (size_t)frac / 8) & ~(size_t)63;
        if (q == 0) break;
        const int ngroups = pow3(R);
        struct Prod { const G1A* a; const G2A* b; int g; };
        std::vector<Prod> prods;
        // B block outermost: the 2^R products that pair one B block with different A blocks are ADJACENT and share its G2 chain (ChainSets)
        for (int fb = 0; fb < (1 << R); ++fb) for (int eb = 0; eb < (1 << R); ++eb) {
            int g = 0;
            for (int t = 0; t < R; ++t) g = g * 3 + (((eb >> (R - 1 - t)) & 1) - ((fb >> (R - 1 - t)) & 1) + 1);
            const size_t ia = ((size_t)eb << 1) | (side == 0 ? 1u : 0u), ib = ((size_t)fb << 1) | (side == 0 ? 0u : 1u);
            prods.push_back({a + ia * qblk, b + ib * qblk, g});
        }
        std::vector<Fp12> grows((size_t)ngroups * N_LINES); std::vector<char> gset((size_t)ngroups, 0);
        const size_t cmax = std::min<size_t>(MAX_PRODUCTS, std::max<size_t>(1, e->pairs_cap(std::min(e->max_pairs_per_batch, q * (size_t)MAX_PRODUCTS)) / q));
        const size_t nch = (prods.size() + cmax - 1) / cmax;
        auto absorb = [&](size_t c) -> int32_t {                 // chunk c's per-step values into their groups (the device is busy with chunk c + 1)
            PinBuf& buf = j->look_rows[c & 1];
            int32_t r2 = buf.wait(); if (r2) return r2;
            const Fp12* rows = buf.as<Fp12>();
            const size_t lo = c * cmax, hi = std::min(prods.size(), lo + cmax);
            for (size_t k = lo; k < hi; ++k) {
                Fp12* dst = grows.data() + (size_t)prods[k].g * N_LINES; const Fp12* src = rows + (k - lo) * N_LINES;
                if (!gset[prods[k].g]) { std::memcpy(dst, src, N_LINES * sizeof(Fp12)); gset[prods[k].g] = 1; }
                else for (int s2 = 0; s2 < N_LINES; ++s2) dst[s2] = mul(dst[s2], src[s2]);
            }
            return RIPP_OK;
        };
        for (size_t c = 0; c < nch; ++c) {
            PinBuf& buf = j->look_rows[c & 1];
            if ((rc = buf.reserve(MAX_PRODUCTS * N_LINES * sizeof(Fp12)))) return rc;
            const size_t lo = c * cmax, hi = std::min(prods.size(), lo + cmax);
            const G1A* as[MAX_PRODUCTS]; const G2A* bs[MAX_PRODUCTS];
            for (size_t k = lo; k < hi; ++k) { as[k - lo] = prods[k].a; bs[k - lo] = prods[k].b; }
            if ((rc = e->enqueue_products(as, bs, (int)(hi - lo), 0, q, buf.as<Fp12>()))) return rc;
            HIPCHK(hipEventRecord(buf.ev, e->stream)); buf.pending = true;
            e->stats.look_pairs += (hi - lo) * q;
            if (c > 0 && (rc = absorb(c - 1))) return rc;
        }
        if ((rc = absorb(nch - 1))) return rc;
        j->look.emplace_back();
        ripp_sipp_job::LookItem& li = j->look.back();
        li.R = R; li.side = side; li.level = 0; li.npairs = q; li.q = qblk;
        for (int g = 0; g < ngroups; ++g) {
            auto rows = std::make_shared<std::vector<Fp12>>(grows.begin() + (size_t)g * N_LINES, grows.begin() + (size_t)(g + 1) * N_LINES);
            li.fe.push_back(look_pool().submit([rows]() { return final_exponentiation(miller_combine(rows->data())); }));
        }
        ++e->stats.look_items;
    }
    e->stats.look_ms += now_ms() - t0;
    if (trace_on()) fprintf(stderr, "[ripp] look-ahead: %zu of %d (round, side) values pre-evaluated in the hash window, %.1f ms (hash %s)\n", j->look.size(), items, now_ms() - t0, j->digest_ready.load() ? "already done" : "still running");
    return RIPP_OK;
}
static void look_finish_level(ripp_sipp_job::LookItem& it) {
    if (it.pend.empty()) return;
    const size_t S = it.Z.size() / 3;
    std::vector<Fp12> Zn(S);
    for (size_t r = 0; r < S; ++r) { const Fp12 lo = mul(it.pend[3 * r].get(), it.pend[3 * r + 1].get()); Zn[r] = mul(mul(lo, it.Z[S + r]), it.pend[3 * r + 2].get()); }
    it.pend.clear(); it.Z.swap(Zn); ++it.level;
}
static void look_start_level(ripp_sipp_job::LookItem& it, const GlsDigits& gx, const GlsDigits& gxi) {
    if (!it.fe.empty()) { it.Z.resize(it.fe.size()); for (size_t i = 0; i < it.fe.size(); ++i) it.Z[i] = it.fe[i].get(); it.fe.clear(); }
    const size_t S = it.Z.size() / 3;
    for (size_t r = 0; r < S; ++r) {          // 1/x is full width (four digit strings: two tasks), x is 128 bits (two strings: one task)
        const Fp12 zm = it.Z[r], zp = it.Z[2 * S + r];
        it.pend.push_back(host_pool().submit([zm, gxi]() { return gt_pow_gls_strings(zm, gxi, 3u); }));
        it.pend.push_back(host_pool().submit([zm, gxi]() { return gt_pow_gls_strings(zm, gxi, 12u); }));
        it.pend.push_back(host_pool().submit([zp, gx]() { return gt_pow_gls_strings(zp, gx, 15u); }));
    }
}
// challenge x_t is known: apply it to every item of a later round -- the next round's first (critical), the deeper ones behind it
static void look_apply(ripp_sipp_job* j, size_t t, const Fr& x) {
    bool any = false; for (auto& it : j->look) any = any || (size_t)it.R > t;
    if (!any) return;
    const GlsDigits gx = gls_digits(x), gxi = gls_digits(inv(x));
    for (int pass = 0; pass < 2; ++pass) for (auto& it : j->look) {
        if ((size_t)it.R <= t || ((size_t)it.R == t + 1) != (pass == 0)) continue;
        look_finish_level(it);
        look_start_level(it, gx, gxi);
    }
}
static ripp_sipp_job::LookItem* look_find(ripp_sipp_job* j, size_t round, int side) { for (auto& it : j->look) if ((size_t)it.R == round && it.side == side) return &it; return nullptr; }
// both values of `round` are known in full: nothing of that round is left for the device
static bool look_full(ripp_sipp_job* j, size_t round) { const auto* l = look_find(j, round, 0); const auto* r = look_find(j, round, 1); return l && r && l->npairs == l->q && r->npairs == r->q; }

// Round 0 and the look-ahead item (1,l) in ONE evaluation.  With the quarters A0..A3 / B0..B3 of the round-0 vectors, round 0 needs
// z_l = E(A2,B0) E(A3,B1), z_r = E(A0,B2) E(A1,B3) and item (1,l) the four products E(A1,B0), E(A1,B2), E(A3,B0), E(A3,B2): B0 and B2 each meet THREE
// A blocks, so their G2 chains are walked once for three line sets (fq_miller.hpp) -- 4 chains instead of 8 for these eight quarter products, and
// item (1,r) pairs B1 and B3 with two A blocks each (job_lookahead): 6 chain walks per quarter where the separate evaluations took 12.
// rows: the per-step products of z_l and z_r like job_round_partials; the item is appended to j->look (exponent groups d = e - f as in job_lookahead).
static int32_t job_round0_shared(Engine* e, ripp_sipp_job* j, Fp12* rows) {
    const size_t q = j->len / 4;
    const G1A* a = j->a.as<G1A>(); const G2A* b = j->b.as<G2A>();
    const G1A* as[8] = {a + 2 * q, a + q, a + 3 * q,   a, a + q, a + 3 * q,   a + 3 * q,   a + q};
    const G2A* bs[8] = {b, b, b,   b + 2 * q, b + 2 * q, b + 2 * q,   b + q,   b + 3 * q};
    std::vector<Fp12> all((size_t)8 * N_LINES);
    const double t0 = now_ms();
    int32_t rc = e->step_products(as, bs, 8, q, all.data());
    e->stats.miller_products_ms += now_ms() - t0;
    if (rc) return rc;
    auto row = [&](int p) { return all.data() + (size_t)p * N_LINES; };
    for (int s2 = 0; s2 < N_LINES; ++s2) { rows[s2] = mul(row(0)[s2], row(6)[s2]); rows[N_LINES + s2] = mul(row(3)[s2], row(7)[s2]); }
    j->look.clear();
    j->look.emplace_back();
    ripp_sipp_job::LookItem& li = j->look.back();
    li.R = 1; li.side = 0; li.level = 0; li.npairs = q; li.q = q;
    // exponent groups g = e - f + 1:  0: E(A1,B2)   1: E(A1,B0) E(A3,B2)   2: E(A3,B0)
    for (int g = 0; g < 3; ++g) {
        auto grows = std::make_shared<std::vector<Fp12>>((size_t)N_LINES);
        for (int s2 = 0; s2 < N_LINES; ++s2) (*grows)[s2] = g == 0 ? row(4)[s2] : g == 1 ? mul(row(1)[s2], row(5)[s2]) : row(2)[s2];
        li.fe.push_back(look_pool().submit([grows]() { return final_exponentiation(miller_combine(grows->data())); }));
    }
    ++e->stats.look_items; e->stats.look_pairs += 4 * q;
    return RIPP_OK;
}

// ---- SIPP::prove (sipp/src/lib.rs:42-106) on this rank's shard; world0 == 1: the whole proof on one GPU --------------------------------------
// One protocol for every world size: the ranks walk the same sequence of exchanges (plan, one per round while the vectors are sharded, the
// tail gather), every message carries the sender's status, and a rank that failed locally keeps walking until the next exchange has told the
// others -- no rank is left blocked in a collective (all ranks return an error together).
// Test hook (ripp_test_inject_failure): the fold of round `round` on rank `rank` of the NEXT sharded / single proof reports a device error instead of
// running -- what an allocation or launch failure in the middle of a proof looks like to the protocol.  One shot: consumed when it fires.
static std::atomic<int> g_fail_rank{-1}, g_fail_round{-1};
static bool test_fail_hit(int rank, size_t round) {
    if (g_fail_rank.load(std::memory_order_relaxed) == rank && g_fail_round.load(std::memory_order_relaxed) == 1000 + (int)round) raise(SIGKILL);      // (round + 1000: the rank DIES there)
    if (g_fail_rank.load(std::memory_order_relaxed) != rank || g_fail_round.load(std::memory_order_relaxed) != (int)round) return false;
    g_fail_rank = -1; g_fail_round = -1;
    set_err("injected failure (ripp_test_inject_failure) in the fold of round " + std::to_string(round) + " on rank " + std::to_string(rank));
    return true;
}
struct SippPlanMsg { uint64_t n_local; int32_t world, rank, look_items, window, rc, pad; double hash_left_ms; };      // hash_left_ms (rank 0): what its statement hash still needs, by its measured rate
struct SippRoundMsg { Fp12 z[2]; uint8_t digest[32]; int32_t rc, pad[3]; };
struct SippTailMsg { G1A a; G2A b; int32_t rc, pad[3]; };
static int32_t sipp_prove_core(Engine* e, ripp_sipp_job* j, const Fp12& val, const uint8_t* seed_digest, ripp_gt* proof, ripp_fr* challenges, ripp_stats* st) {
    const int world0 = j->world0, rank = j->rank;
    const double t_start = now_ms();
    double exchange_ms = 0;
    comm_mark_proof();
    e->mem_tier = 0;
    // whatever the exit path -- the plan exchange below included: nothing enqueued by this proof may still be running when the caller gets control
    // back (engine scratch, tp_rows and the job's vectors are reused by the next call), no prepared state may leak into the next proof, and the hash
    // thread, which reads the CALLER's buffers (borrowed statement), is never left running behind a return.  Constructed BEFORE the thread starts.
    struct Quiesce { Engine* e; ripp_sipp_job* j; ~Quiesce() {
        (void)hipStreamSynchronize(e->stream); (void)hipStreamSynchronize(e->stream2); (void)hipStreamSynchronize(e->stream3);
        j->tp_round[0] = j->tp_round[1] = ~(size_t)0; j->pre_vm_ready = false; j->pre_vm_side = false; j->pre_ready = false; j->tab_ready = false; j->look.clear();
        e->g2tab_hi = nullptr;
        if (j->hash_thread.joinable()) j->hash_thread.join();
        j->ha_ext = nullptr; j->hb_ext = nullptr; j->hr_ext = nullptr; j->hash_prestarted = false; } } quiesce{e, j};
    // rank 0 (the only rank of a single-GPU proof) hashes the statement on a host thread: THE serial floor, started before anything else
    const bool window = rank == 0 && !seed_digest;
    if (rank == 0) {
        if (seed_digest) { if (j->hash_thread.joinable()) j->hash_thread.join(); std::memcpy(j->digest, seed_digest, 32); j->digest_ready = true; j->hash_prestarted = false; }
        else job_start_hash(j, val);
    }
    const bool look_forced = e->look_eighths >= 0;
    int look_items = (rank == 0 || look_forced) ? look_plan(e, j->n_local, world0, window || look_forced) : 0;
    j->no_window = !window; j->look_deadline = 0;
    if (world0 > 1) {
        double hash_left = 0;
        if (window && j->hash_total > 0) hash_left = std::max(0.0, (double)j->hash_total / (e->cal_hash_bytes_per_ms > 0 ? e->cal_hash_bytes_per_ms : 1.17e6) - (now_ms() - j->hash_t0));
        SippPlanMsg mine{(uint64_t)j->n_local, world0, rank, look_items, window ? 1 : 0, RIPP_OK, 0, hash_left};
        std::vector<SippPlanMsg> all((size_t)world0);
        const double tx = now_ms();
        t_ex_kind = EX_PLAN;
        int32_t rc = comm_allgather(e, &mine, all.data(), sizeof mine); if (rc) return rc;
        exchange_ms += now_ms() - tx;
        for (int w = 0; w < world0; ++w) {
            if (all[w].rc) { set_err("sharded SIPP proof: rank " + std::to_string(w) + " failed before the proof started (status " + std::to_string(all[w].rc) + ")"); return RIPP_ERR_DEVICE; }
            if (all[w].n_local != (uint64_t)j->n_local || all[w].world != world0 || all[w].rank != w) { set_err("sharded SIPP proof: the ranks disagree on the shard size / world size / rank order"); return RIPP_ERR_ARG; }
        }
        look_items = all[0].look_items; j->no_window = !all[0].window;
        if (rank != 0 && all[0].window && all[0].hash_left_ms > 0) j->look_deadline = now_ms() + all[0].hash_left_ms;      // (durations, not clocks: every rank leaves the all-gather at about the same time)
    }
    struct HotOff { ~HotOff() { host_pool().set_hot(false); } } hot_off;       // whatever the exit path, the workers go back to sleeping waits
    struct QuietOff { Engine* e; ~QuietOff() { e->quiet_waits = false; } } quiet_off{e};
    // RIPP_QUIET_WAITS=1: sleeping instead of spinning waits while this rank hashes.  Measured A/B on four boxes (profiles/r03_quiet_vs_spin_waits.txt):
    // the hash is not faster for it (313-322 vs 312-330 ms) and the wake-up latencies make the adaptive look-ahead overrun the window: 460-470 ms
    // against 458-465 ms with spinning waits.  Off by default.
    e->quiet_waits = window && e->quiet_waits_cfg;
    j->look_rows[0].blocking = j->look_rows[1].blocking = e->quiet_waits;
    struct XsOff { ripp_sipp_job* j; ~XsOff() { j->xs_enabled = false; } } xs_off{j};
    j->xs_enabled = true; j->seeded = false;
    int32_t lrc = job_begin(e, j);                                        // local status: carried to the next exchange while the proof is sharded
    if (lrc && world0 == 1) return lrc;
    e->stats.look_items = 0; e->stats.look_pairs = 0; e->stats.look_ms = 0;
    if (trace_on()) fprintf(stderr, "[ripp] scale+normalize done at t=%.1f ms\n", now_ms() - t_start);
    size_t len = j->n_local;                                              // protocol view of the local length (j->len follows it unless a local step failed)
    bool sharded = world0 > 1, digest_sent = world0 == 1;
    size_t round = 0;
    Fr x_prev = Fr::zero();
    // fused fold of rounds 0 and 1 (job_fold_fused): with both values of round 1 known from the look-ahead, round 0's fold is DEFERRED until x1 exists
    // (the GT powers of the look-ahead values and one hash later, ~1 ms) and the vectors go from n to n / 4 in one pass over the three-quarter tables
    bool fuse_pending = false; Fr x0_saved = Fr::zero();
    j->tp_round[0] = j->tp_round[1] = ~(size_t)0;
    while ((sharded ? len * (size_t)world0 : len) > 1) {
        if (sharded && len == 1) {
            // tail: every rank holds ONE element; gather them (rank order == global order) and finish replicated (sipp/src/lib.rs:69-104 on world0 elements)
            SippTailMsg mine; std::memset((void*)&mine, 0, sizeof mine);
            if (!lrc) {
                auto grab = [&]() -> int32_t {
                    HIPCHK(hipStreamSynchronize(e->stream2)); HIPCHK(hipStreamSynchronize(e->stream3));
                    HIPCHK(hipMemcpyAsync(&mine.a, j->a.p, sizeof(G1A), hipMemcpyDeviceToHost, e->stream));
                    HIPCHK(hipMemcpyAsync(&mine.b, j->b.p, sizeof(G2A), hipMemcpyDeviceToHost, e->stream));
                    return e->sync(); };
                lrc = grab();
            }
            mine.rc = lrc;
            std::vector<SippTailMsg> all((size_t)world0);
            const double tx = now_ms();
            int32_t rc = comm_allgather(e, &mine, all.data(), sizeof mine); if (rc) return rc;
            exchange_ms += now_ms() - tx;
            for (int w = 0; w < world0; ++w) if (all[w].rc) { if (!lrc) set_err("sharded SIPP proof: rank " + std::to_string(w) + " failed (status " + std::to_string(all[w].rc) + ")"); return lrc ? lrc : RIPP_ERR_DEVICE; }
            const size_t L = (size_t)world0;
            if ((rc = j->a.reserve(L * sizeof(G1A))) || (rc = j->b.reserve(L * sizeof(G2A))) || (rc = j->a_next.reserve(L * sizeof(G1A))) ||
                (rc = j->b_next.reserve(L * sizeof(G2A))) || (rc = j->jac1.reserve(L * sizeof(G1J))) || (rc = j->jac2.reserve(L * sizeof(G2J))) || (rc = e->stage[3].reserve(L * (sizeof(G1A) + sizeof(G2A))))) return rc;
            G1A* ha = e->stage[3].as<G1A>(); G2A* hb = reinterpret_cast<G2A*>(ha + L);
            for (size_t w = 0; w < L; ++w) { ha[w] = all[w].a; hb[w] = all[w].b; }
            if (j->bs_on) { set_err("sharded SIPP proof: the G2 vector is still x-scaled at the tail gather"); return RIPP_ERR_ARG; }      // (cannot happen: small rounds fold the plain vector)
            HIPCHK(hipMemcpyAsync(j->a.p, ha, L * sizeof(G1A), hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(j->b.p, hb, L * sizeof(G2A), hipMemcpyHostToDevice, e->stream));
            if ((rc = e->sync())) return rc;
            j->len = len = L; j->world = 1; sharded = false;
            j->pre_vm_ready = false;
            continue;
        }
        Fp12 rows[2 * N_LINES];
        const double tr0 = now_ms();
        ripp_sipp_job::LookItem* lk_l = lrc ? nullptr : look_find(j, round, 0);
        ripp_sipp_job::LookItem* lk_r = lrc ? nullptr : look_find(j, round, 1);
        const bool tp_round = !lrc && j->tp_round[round & 1] == round;                       // pipelined tail: this round's products were enqueued a round ago
        double t0 = tr0;
        double round0_ms_per_pair = 7.7e-5;                                                     // measured below in round 0 (lines + products + tree + copy per pair)
        bool shared_r0 = false;                                                                 // round 0 evaluated together with look-ahead item (1,l)
        Fp12 zl = Fp12::one(), zr = Fp12::one();
        auto local_values = [&]() -> int32_t {
            int32_t rc;
            // what the look-ahead has not covered goes to the device: pairs [start_s, half) of z_l = prod e(a_r, b_l) (s = 0) and z_r = prod e(a_l, b_r) (s = 1)
            const size_t half = fuse_pending ? j->len / 4 : j->len / 2;       // (fuse_pending: the job's vectors are still round 0's; both values of this round come from the look-ahead)
            const size_t start[2] = {lk_l ? std::min(lk_l->npairs, half) : 0, lk_r ? std::min(lk_r->npairs, half) : 0};
            const bool dev[2] = {!tp_round && start[0] < half, !tp_round && start[1] < half};
            if (dev[0] && dev[1] && start[0] == 0 && start[1] == 0) {
                const double tp = now_ms();
                // round 0 with item (1,l) of the look-ahead planned in full: one evaluation with shared G2 chains (job_round0_shared)
                shared_r0 = round == 0 && !j->seeded && look_items >= 8 && j->len >= 4 && !e->sw.no_share && !e->sw.no_precompute && (look_forced || !j->digest_ready.load());
                if (shared_r0) { if ((rc = job_round0_shared(e, j, rows))) return rc; }
                else if ((rc = job_round_partials(e, j, rows))) return rc;
                if (round == 0 && j->len >= ((size_t)1 << 16)) {
                    round0_ms_per_pair = (now_ms() - tp) / (double)(shared_r0 ? 2 * j->len : j->len);      // 2 products x len / 2 pairs (shared: 8 x len / 4)
                    if (e->ranks_per_device <= 1.0) e->cal_ms_per_pair = round0_ms_per_pair;                 // (a time-sliced device would count its neighbours' work)
                }
            }
            else for (int sd = 0; sd < 2; ++sd) {
                if (!dev[sd]) continue;
                const G1A* as[1] = {j->a.as<G1A>() + (sd == 0 ? half : 0) + start[sd]}; const G2A* bs[1] = {j->b.as<G2A>() + (sd == 0 ? 0 : half) + start[sd]};
                const double tp = now_ms();
                if ((rc = e->step_products(as, bs, 1, half - start[sd], rows + sd * N_LINES))) return rc;
                e->stats.miller_products_ms += now_ms() - tp;
            }
            // asynchronous, during the host work below: the challenge-independent half of this round's G2 table fold (rounds of 2^16 and more elements)
            if (round >= 1 && !fuse_pending && !tp_round && (rc = job_prebuild_g2_tables(e, j))) return rc;
            // asynchronous: overlaps the host work below and the hash.  Both items of round 1 planned in full: tables over three quarters for the fused fold
            // Three-quarter tables (fused fold of rounds 0 + 1) as soon as the plan expects at least HALF of item (1,r) to fit: the adaptive planner completes
            // that item when >= 80 % of it fits (job_lookahead), and a forced full plan measured 409 ms against 416-418 ms for the cautious one on a 289 ms-hash
            // box (profiles/r05_plan_threshold_ab.txt) -- since the x86-64 Blake2s loop the window is ~20 ms shorter than the work that used to fill it.
            const bool plan_fuse = look_forced ? look_items >= 16 : look_items >= 12;
            if (round == 0 && !j->seeded && (rc = job_precompute_round0(e, j, (plan_fuse || e->sw.fuse_tables) && j->len >= 4))) return rc;
            if (!j->pre_vm_ready && (rc = job_precompute_vm(e, j))) return rc;                   // small rounds: the same on the VM, during the host phase
            // entry into the pipelined tail -- unless the look-ahead has (round 0: is about to get) both values of the next round
            const bool next_known = round == 0 ? (plan_fuse && j->len >= 4 && (look_forced || !j->digest_ready.load())) : look_full(j, round + 1);
            if (!tp_round && !next_known && tail_pipe_ok(e, j) && (rc = job_tail_enqueue(e, j, round + 1))) return rc;
            t0 = now_ms();
            if (tp_round) { if ((rc = job_tail_values(j, round, x_prev, &zl, &zr))) return rc; }
            else {
                Fp12 zd[2] = {Fp12::one(), Fp12::one()};                              // the device's share of each value
                if (dev[0] && dev[1]) pairing_values(rows, 2, zd);
                else if (dev[0]) pairing_values(rows, 1, &zd[0]);
                else if (dev[1]) pairing_values(rows + N_LINES, 1, &zd[1]);
                if (j->bs_on && (dev[0] || dev[1])) {      // the device holds bs * b: what it evaluated is z^bs (look-ahead values came from the plain round-0 blocks)
                    const GlsDigits gsi = gls_digits(inv(j->bs));                     // two tasks of two digit strings per value
                    if (host_pool().size() >= 11) {                                     // one digit string per task
                        Fp12 part[8];
                        host_pool().parallel(8, [&](int t) { if (!dev[t >> 2]) return; part[t] = gt_pow_gls_strings(zd[t >> 2], gsi, 1u << (t & 3)); });
                        for (int k = 0; k < 2; ++k) if (dev[k]) zd[k] = mul(mul(part[4 * k], part[4 * k + 1]), mul(part[4 * k + 2], part[4 * k + 3]));
                    } else {
                    Fp12 part[4];
                    host_pool().parallel(4, [&](int t) { if (!dev[t >> 1]) return; part[t] = gt_pow_gls_strings(zd[t >> 1], gsi, (t & 1) ? 12u : 3u); });
                    for (int k = 0; k < 2; ++k) if (dev[k]) zd[k] = mul(part[2 * k], part[2 * k + 1]);
                    }
                }
                zl = zd[0]; zr = zd[1];
                if (lk_l) { look_finish_level(*lk_l); zl = dev[0] ? mul(lk_l->Z[0], zd[0]) : lk_l->Z[0]; }
                if (lk_r) { look_finish_level(*lk_r); zr = dev[1] ? mul(lk_r->Z[0], zd[1]) : lk_r->Z[0]; }
            }
            if (round == 0 && !j->seeded && (rc = job_lookahead(e, j, look_items, look_forced, round0_ms_per_pair, shared_r0 ? 1 : 0))) return rc;      // blocks on the GPU while the hash thread is still busy
            return RIPP_OK;
        };
        if (!lrc) lrc = local_values();
        if (lrc && !sharded) return lrc;
        if (!j->seeded && rank == 0) {
            const double th = now_ms();
            if (j->hash_thread.joinable()) j->hash_thread.join();
            e->stats.hash_ms += now_ms() - th;               // time the prover actually WAITED for the statement hash
        }
        if (sharded || !digest_sent) {
            // the ranks' values multiply to the whole one (Miller recurrence, final exponentiation, GT powers: all homomorphisms); rank 0's first
            // message also carries the digest of the statement
            SippRoundMsg mine; std::memset((void*)&mine, 0, sizeof mine);
            mine.z[0] = zl; mine.z[1] = zr; mine.rc = lrc;
            if (rank == 0) std::memcpy(mine.digest, j->digest, 32);
            std::vector<SippRoundMsg> all((size_t)world0);
            const double tx = now_ms();
            int32_t rc = comm_allgather(e, &mine, all.data(), sizeof mine); if (rc) return rc;
            exchange_ms += now_ms() - tx;
            for (int w = 0; w < world0; ++w) if (all[w].rc) { if (!lrc) set_err("sharded SIPP proof: rank " + std::to_string(w) + " failed (status " + std::to_string(all[w].rc) + ")"); return lrc ? lrc : RIPP_ERR_DEVICE; }
            if (sharded) { zl = all[0].z[0]; zr = all[0].z[1]; for (int w = 1; w < world0; ++w) { zl = mul(zl, all[w].z[0]); zr = mul(zr, all[w].z[1]); } }
            if (!digest_sent) { std::memcpy(j->digest, all[0].digest, 32); digest_sent = true; }
        }
        if (!j->seeded) {
            j->rng.from_digest(j->digest); j->seeded = true;
            e->quiet_waits = false;
            // polling workers need cores of their own: forced on (ripp_config.hot_workers = 1) with fewer CPUs than workers they would starve the threads that do the work
            // (measured with recorded peers: rank 0 of 8 confined to 4 cores with polling forced: 349 ms after the digest instead of ~50)
            host_pool().set_hot(e->hot_workers_cfg ? (e->hot_workers_cfg == 1 && effective_cpus() >= (int)host_pool().size() + 2) : hot_workers_pay(world0));      // the remaining rounds hand 0.1-0.6 ms tasks to the workers every ~2 ms
        }
        const Fr x = fs::sipp_challenge(j->rng, zl, zr);
        x_prev = x;
        look_apply(j, round, x);                             // GT powers of the pre-evaluated rounds start on the workers right away
        e->stats.host_ms += now_ms() - t0;
        std::memcpy(&proof[2 * round], &zl, sizeof(Fp12)); std::memcpy(&proof[2 * round + 1], &zr, sizeof(Fp12));
        if (challenges) std::memcpy(&challenges[round], &x, sizeof(Fr));
        const double tf0 = now_ms();
        // nobody waits for this fold when the next round's values are already on their way (pipelined tail) or known (look-ahead)
        const bool pipelined = j->tp_round[(round + 1) & 1] == round + 1;
        const bool known = look_full(j, round + 1);
        if (!sharded && len == 2) { j->len = 0; j->pre_vm_ready = false; }      // the one-element vectors of the LAST fold are discarded by the prover (sipp/src/lib.rs:87-104 ends the loop): not computed
        else {
            if (!lrc && test_fail_hit(rank, round)) lrc = RIPP_ERR_DEVICE;
            if (!lrc && fuse_pending) { lrc = job_fold_fused(e, j, x0_saved, x, pipelined || known); fuse_pending = false; }
            else if (!lrc && round == 0 && known && j->tab_ready && j->tab_fused && e->tab_owner == j && !j->bs_on) { fuse_pending = true; x0_saved = x; }      // fold with round 1's
            else
            if (!lrc) lrc = job_fold(e, j, x, true, pipelined || known);
            if (!lrc && pipelined) {              // behind the fold, on the new vectors: the second fold bases of the next round, the values of the round after it
                if (!j->pre_vm_ready) lrc = job_precompute_vm(e, j, true);
                if (!lrc && tail_pipe_ok(e, j)) lrc = job_tail_enqueue(e, j, round + 2);
            }
            if (lrc && !sharded) return lrc;
        }
        len /= 2;
        if (trace_on()) fprintf(stderr, "[ripp] round %2zu len %8zu: products %.2f ms, host %.2f ms, fold %.2f ms%s (t=%.1f)\n", round, len * 2, t0 - tr0, tf0 - t0, now_ms() - tf0, (pipelined || known) ? " (enqueued)" : "", now_ms() - t_start);
        ++round;
    }
    if (j->hash_thread.joinable()) j->hash_thread.join();     // n == 1: no rounds
    host_pool().set_hot(false);
    if (lrc) return lrc;
    int32_t rc;
    if ((rc = e->sync())) return rc; HIPCHK(hipStreamSynchronize(e->stream2)); HIPCHK(hipStreamSynchronize(e->stream3));      // the pipelined tail does not wait for its folds
    e->collect_kernel_stats();
    e->stats.exchange_ms = exchange_ms;
    e->stats.mem_tier = (uint64_t)e->mem_tier; e->stats.device_bytes = g_dev_bytes.load();
    e->stats.statement_hash_ms = window ? g_digest_hash_ms : 0; e->stats.statement_hash_wait_ms = window ? g_digest_wait_ms : 0;      // this call's hash (the thread has been joined)
    if (window && j->hash_total >= ((uint64_t)336 << 16) && g_digest_hash_ms + g_digest_wait_ms > 1.0) e->cal_hash_bytes_per_ms = (double)j->hash_total / (g_digest_hash_ms + g_digest_wait_ms);
    e->stats.total_ms = now_ms() - t_start;
    if (st) *st = e->stats;
    return RIPP_OK;
}

// ---- configuration (include/ripp_hip.h: ripp_config) ------------------------------------------------------------------------------------------
static void config_from_engine(const Engine* e, ripp_config* c) {
    std::memset(c, 0, sizeof *c); c->struct_size = (uint32_t)sizeof *c;
    c->no_vm = e->sw.no_vm; c->no_precompute = e->sw.no_precompute; c->no_fold_tables = e->sw.no_fold_tables; c->no_msm_glv = e->sw.no_msm_glv; c->lp_one_lane = e->sw.lp_one_lane;
    c->no_endo = e->sw.no_endo; c->no_fq = e->sw.no_fq; c->no_xscale = e->sw.no_xscale; c->no_share = e->sw.no_share; c->no_fuse = e->sw.no_fuse; c->fuse_tables = e->sw.fuse_tables; c->scale_no_fq = e->scale_no_fq; c->agg_sequential = e->agg_sequential; c->look_static = e->look_static; c->quiet_waits = e->quiet_waits_cfg;
    c->look_eighths = e->look_eighths; c->ranks_per_device = (int32_t)e->ranks_per_device; c->msm_c = e->msm_tune.c; c->msm_ch = e->msm_tune.ch; c->msm_gmin = e->msm_tune.gmin; c->no_prebuild = e->sw.no_prebuild;
    c->vm_lines_max = e->vm_lines_max; c->vm_fold_max = e->vm_fold_max; c->vm_tree_max = e->vm_tree_max; c->gls_split_max = e->gls_split_max; c->msm_vm_merge_max = e->msm_vm_merge_max; c->fold_tab_min = e->fold_tab_min;
    c->fq_min = e->fq_min; c->lp_fq_min = e->lp_fq_min; c->vm_joint_max = e->vm_joint_max; c->vm_scale_max = e->vm_scale_max; c->tail_pipe_max = e->tail_pipe_max; c->ml_fq_min = e->ml_fq_min; c->fq_min_g1 = e->fq_min_g1; c->msm_lds_sort_min = e->msm_lds_sort_min; c->msm_chunk_min = e->msm_chunk_min;
    c->mem_cap_bytes = e->mem_cap; c->hot_workers = (uint32_t)e->hot_workers_cfg; c->no_job_cache = e->sw.no_job_cache;
    c->no_lp_karatsuba = e->sw.no_lp_kara; c->comm_timeout_ms = e->comm_timeout_ms; c->plan_derate_pct = e->plan_derate_pct; c->n_devices = e->n_devices_cfg;
}
API int32_t ripp_config_default(ripp_config* cfg) {            // the built-in defaults of this build (needs no device: a throw-away Engine object is never initialised)
    if (!cfg) return RIPP_ERR_ARG;
    Engine tmp; tmp.defaults = Engine::Sizes{tmp.vm_lines_max, tmp.vm_fold_max, tmp.vm_tree_max, tmp.gls_split_max, tmp.msm_vm_merge_max, tmp.fold_tab_min, tmp.fq_min, tmp.lp_fq_min, tmp.vm_joint_max, tmp.vm_scale_max, tmp.tail_pipe_max, tmp.ml_fq_min, tmp.fq_min_g1, tmp.msm_lds_sort_min, tmp.msm_chunk_min};
    config_from_engine(&tmp, cfg); cfg->look_eighths = -1; cfg->ranks_per_device = 1;
    return RIPP_OK;
}
API int32_t ripp_configure(const ripp_config* cfg) {
    LOCK;
    if (!cfg) { g_cfg_set = false; return RIPP_OK; }
    if (cfg->struct_size != sizeof(ripp_config)) { set_err("ripp_configure: struct_size does not match this library's ripp_config (fill the struct with ripp_config_default first)"); return RIPP_ERR_ARG; }
    g_cfg = *cfg; g_cfg_set = true;
    return RIPP_OK;
}
API int32_t ripp_config_get(ripp_config* cfg) {                // what the NEXT call runs with: defaults < ripp_configure < environment
    if (!cfg) return RIPP_ERR_ARG;
    LOCK; ENGINE;
    config_from_engine(e, cfg);
    return RIPP_OK;
}

API void ripp_test_inject_failure(int32_t rank, int32_t round) { g_fail_rank = rank; g_fail_round = round; }

API int32_t ripp_sipp_job_prove(ripp_sipp_job* j, const ripp_gt* value, ripp_gt* proof, ripp_fr* challenges, ripp_stats* st) {
    LOCK; ENGINE; if (!j || !value || !proof || j->world0 != 1) return RIPP_ERR_ARG;
    Fp12 val; std::memcpy(&val, value, sizeof(Fp12));
    return sipp_prove_core(e, j, val, nullptr, proof, challenges, st);
}

// SIPP::prove on D IN-PROCESS RANKS (ripp_config.n_devices = D > 1): one host thread, engine and job per device walk the sharded protocol of sipp_prove_core side by
// side -- the statement sharded by index residue exactly as across processes, rank 0 hashing the caller's whole statement, the per-round exchange of finished
// partial GT values a copy through LocalComm (comm_core.inc).  Every rank ends with the same proof; the caller gets rank 0's after they have been compared.
// The unmodified single-process caller of SIPP::prove thereby reaches every device it drives.  UNMEASURED on a multi-GPU node; tested on RIPP_VIRTUAL_DEVICES.
extern "C++" {
static int32_t sipp_prove_devices(Engine* e, std::vector<DevSlot>& sl, const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n, const ripp_gt* value,
                                  ripp_gt* proof, ripp_fr* challenges, ripp_stats* st) {
    const size_t D = sl.size(), nl = n / D; size_t rounds = 0; while (((size_t)1 << rounds) < n) ++rounds;
    LocalComm lc; lc.world = (int)D;
    Fp12 val; std::memcpy(&val, value, sizeof val);
    struct Out { std::vector<ripp_gt> proof; std::vector<ripp_fr> ch; ripp_stats st; };
    std::vector<Out> outs(D);
    const bool virt = e->virtual_devices;
    int32_t rc = run_on_devices(e, sl, [&](Engine* ee, size_t, size_t, int d) -> int32_t {
        t_lcomm = &lc; t_lrank = d;
        struct Unbind { ~Unbind() { t_lcomm = nullptr; t_lrank = 0; } } unbind;
        if (virt) ee->ranks_per_device = (double)D;                              // the look-ahead plan prices the window per DEVICE
        std::vector<G1A> sa(nl); std::vector<G2A> sb(nl); std::vector<Fr> sr(nl);      // this rank's residue class (local j <-> global j D + d)
        for (size_t i = 0; i < nl; ++i) { std::memcpy(&sa[i], &a[i * D + (size_t)d], sizeof(G1A)); std::memcpy(&sb[i], &b[i * D + (size_t)d], sizeof(G2A)); std::memcpy(&sr[i], &r[i * D + (size_t)d], sizeof(Fr)); }
        ripp_sipp_job* j = nullptr;
        int32_t r2 = sipp_job_create_on(ee, reinterpret_cast<const ripp_g1a*>(sa.data()), reinterpret_cast<const ripp_g2a*>(sb.data()), reinterpret_cast<const ripp_fr*>(sr.data()), nl, d, (int32_t)D,
                                        d == 0 ? value : nullptr, &j, d == 0 ? a : nullptr, d == 0 ? b : nullptr, d == 0 ? r : nullptr, n, true);
        if (r2) { lc.fail(); return r2; }
        if (d == 0 && !j->hash_prestarted) {                                    // the caller's buffers do not meet the engine types' alignment: hash a copy
            j->ha_ext = nullptr; j->ha.resize(n); j->hb.resize(n); j->hr.resize(n);
            std::memcpy(j->ha.data(), a, n * sizeof(G1A)); std::memcpy(j->hb.data(), b, n * sizeof(G2A)); std::memcpy(j->hr.data(), r, n * sizeof(Fr));
        }
        Out& o = outs[(size_t)d]; o.proof.resize(2 * std::max<size_t>(rounds, 1)); o.ch.resize(std::max<size_t>(rounds, 1));
        r2 = sipp_prove_core(ee, j, val, nullptr, o.proof.data(), o.ch.data(), &o.st);
        sipp_job_retire_on(ee, j);
        if (r2) lc.fail();
        return r2; });
    if (rc) return rc;
    for (size_t d = 1; d < D; ++d)
        if (std::memcmp(outs[d].proof.data(), outs[0].proof.data(), 2 * rounds * sizeof(ripp_gt))) { set_err("in-process ranks: the devices' proofs differ"); return RIPP_ERR_DEVICE; }
    std::memcpy(proof, outs[0].proof.data(), 2 * rounds * sizeof(ripp_gt));
    if (challenges) std::memcpy(challenges, outs[0].ch.data(), rounds * sizeof(ripp_fr));
    if (st) *st = outs[0].st;
    return RIPP_OK;
}
}
API int32_t ripp_sipp_prove(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n, const ripp_gt* value, ripp_gt* proof, ripp_fr* challenges, ripp_stats* st) {
    if (n == 0 || (n & (n - 1))) return RIPP_ERR_POW2;
    if (!value) return RIPP_ERR_ARG;
    {   // more than one device configured for this process, and a statement worth sharding (>= 2^14 elements per device)?
        LOCK; ENGINE;
        if (e->n_devices_cfg > 1 && a && b && r && proof) {
            std::vector<DevSlot> sl; int32_t rc = device_slots(e, n, (size_t)1 << 14, &sl); if (rc) return rc;
            size_t D = 1; while (2 * D <= sl.size()) D *= 2;
            g_last_slots = (int32_t)D;
            if (D > 1) { sl.resize(D); return sipp_prove_devices(e, sl, a, b, r, n, value, proof, challenges, st); }
        }
    }
    ripp_sipp_job* j = nullptr;
    int32_t rc = sipp_job_create_impl(a, b, r, n, 0, 1, value, &j, nullptr, nullptr, nullptr, 0, true); if (rc) return rc;
    rc = ripp_sipp_job_prove(j, value, proof, challenges, st);
    sipp_job_retire(j);
    return rc;
}

API int32_t ripp_sipp_seed_digest(const ripp_g1a* a, const ripp_g2a* b, const ripp_fr* r, size_t n, const ripp_gt* value, uint8_t digest[32]) {
    if (!value || !digest || (n && (!a || !b || !r))) return RIPP_ERR_ARG;
    Fp12 v; std::memcpy(&v, value, sizeof(Fp12));
    // the flat C-ABI structs ARE the engine's types (static_asserts at the top of this file): hash in place -- no 336 MB copy on the
    // critical path -- whenever the caller's buffers meet the engine types' 16-byte alignment
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)r) & 15u) == 0) {
        statement_digest(reinterpret_cast<const G1A*>(a), reinterpret_cast<const G2A*>(b), reinterpret_cast<const Fr*>(r), n, v, digest);
        return RIPP_OK;
    }
    std::vector<G1A> ha(n); std::vector<G2A> hb(n); std::vector<Fr> hr(n);
    if (n) { std::memcpy(ha.data(), a, n * sizeof(G1A)); std::memcpy(hb.data(), b, n * sizeof(G2A)); std::memcpy(hr.data(), r, n * sizeof(Fr)); }
    statement_digest(ha.data(), hb.data(), hr.data(), n, v, digest);
    return RIPP_OK;
}

// ---- host helpers ----------------------------------------------------------------------------------------------------
// BLAKE2s-256 of a host buffer (the digest of sipp/src/rng.rs:14; RFC 7693) -- the library's own implementation incl. its x86-64 assembly bulk path, exposed for tests
API int32_t ripp_blake2s(const uint8_t* in, size_t len, uint8_t out[32]) {
    if ((!in && len) || !out) return RIPP_ERR_ARG;
    fs::Blake2s h; h.update(in, len); h.finish(out); return RIPP_OK;
}
API int32_t ripp_final_exp(const ripp_gt* f, ripp_gt* out) { if (!f || !out) return RIPP_ERR_ARG; Fp12 x; std::memcpy(&x, f, sizeof x); const Fp12 r = final_exponentiation(x); std::memcpy(out, &r, sizeof r); return RIPP_OK; }
API int32_t ripp_pairing_values(const ripp_gt* rows, int32_t count, int32_t parts, ripp_gt* out) {
    if (!rows || !out || count < 0 || parts < 0 || parts > 63) return RIPP_ERR_ARG;
    std::vector<Fp12> L((size_t)count * N_LINES), z((size_t)count);
    std::memcpy(L.data(), rows, L.size() * sizeof(Fp12));
    pairing_values(L.data(), count, z.data(), parts);
    std::memcpy(out, z.data(), z.size() * sizeof(Fp12));
    return RIPP_OK;
}
API int32_t ripp_miller_combine(const ripp_gt* rows, ripp_gt* out) { if (!rows || !out) return RIPP_ERR_ARG; std::vector<Fp12> L(N_LINES); std::memcpy(L.data(), rows, N_LINES * sizeof(Fp12)); const Fp12 r = miller_combine(L.data()); std::memcpy(out, &r, sizeof r); return RIPP_OK; }
// sum of a few projective points on the host (the cross-rank reduction of sharded MSM partials: G-1 additions)
API int32_t ripp_sum_g1_j(const ripp_g1j* pts, size_t n, ripp_g1j* out) { if (!out || (n && !pts)) return RIPP_ERR_ARG; G1J acc = jac_inf<Fp>(); for (size_t i = 0; i < n; ++i) { G1J p; std::memcpy(&p, &pts[i], sizeof p); acc = add(acc, p); } std::memcpy(out, &acc, sizeof acc); return RIPP_OK; }
API int32_t ripp_sum_g2_j(const ripp_g2j* pts, size_t n, ripp_g2j* out) { if (!out || (n && !pts)) return RIPP_ERR_ARG; G2J acc = jac_inf<Fp2>(); for (size_t i = 0; i < n; ++i) { G2J p; std::memcpy(&p, &pts[i], sizeof p); acc = add(acc, p); } std::memcpy(out, &acc, sizeof acc); return RIPP_OK; }
API int32_t ripp_gt_mul(const ripp_gt* a, const ripp_gt* b, ripp_gt* out) { if (!a || !b || !out) return RIPP_ERR_ARG; Fp12 x, y; std::memcpy(&x, a, sizeof x); std::memcpy(&y, b, sizeof y); const Fp12 r = mul(x, y); std::memcpy(out, &r, sizeof r); return RIPP_OK; }
API int32_t ripp_gt_pow(const ripp_gt* a, const ripp_fr* k, ripp_gt* out) {
    if (!a || !k || !out) return RIPP_ERR_ARG; Fp12 x; Fr km; std::memcpy(&x, a, sizeof x); std::memcpy(&km, k, sizeof km);
    const Fr c = from_mont(km); Fp12 acc = Fp12::one();
    for (int i = 255; i >= 0; --i) { acc = sqr(acc); if ((c.l[i >> 5] >> (i & 31)) & 1u) acc = mul(acc, x); }
    std::memcpy(out, &acc, sizeof acc); return RIPP_OK;
}
API int32_t ripp_fr_inverse(const ripp_fr* a, ripp_fr* out) { if (!a || !out) return RIPP_ERR_ARG; Fr x; std::memcpy(&x, a, sizeof x); const Fr r = inv(x); std::memcpy(out, &r, sizeof r); return RIPP_OK; }
API size_t ripp_ser_gt(const ripp_gt* f, uint8_t out[576]) { Fp12 x; std::memcpy(&x, f, sizeof x); fs::ser_gt(x, out); return 576; }
API size_t ripp_ser_g1(const ripp_g1a* p, uint8_t out[96]) { G1A x; std::memcpy(&x, p, sizeof x); fs::ser_g1(x, out); return 96; }
API size_t ripp_ser_g2(const ripp_g2a* p, uint8_t out[192]) { G2A x; std::memcpy(&x, p, sizeof x); fs::ser_g2(x, out); return 192; }
API size_t ripp_ser_fr(const ripp_fr* s, uint8_t out[32]) { Fr x; std::memcpy(&x, s, sizeof x); fs::ser_fr(x, out); return 32; }
API int32_t ripp_sipp_challenge(uint8_t seed[32], const ripp_gt* z_l, const ripp_gt* z_r, ripp_fr* x) {
    if (!seed || !z_l || !z_r || !x) return RIPP_ERR_ARG;
    fs::FiatShamirRng rng; rng.from_digest(seed); Fp12 a, b; std::memcpy(&a, z_l, sizeof a); std::memcpy(&b, z_r, sizeof b);
    const Fr c = fs::sipp_challenge(rng, a, b); std::memcpy(seed, rng.seed, 32); std::memcpy(x, &c, sizeof c); return RIPP_OK;
}

#include "vec_api.inc"       // device-resident vectors (ripp_vec_*)

#include "wire_api.inc"      // CanonicalSerialize / CanonicalDeserialize images of the proof structs (zcash layout on BLS12-381, generic SWFlags layout on BLS12-377: wire.hpp)

#include "comm_api.inc"      // RCCL / callback communicator, sharded inner products and the sharded SIPP prover

// ---- synthetic inputs ---------------------------------------------------------------------------------------------------
extern "C++" {
template <class F> static int32_t synth_points(const Affine<F>& g, uint64_t start, size_t first, size_t stride, size_t n, void* out) {
    LOCK; ENGINE; if (n == 0) return RIPP_OK; if (!out) return RIPP_ERR_ARG;
    DevBuf& jac = std::is_same<F, Fp>::value ? e->jacG1 : e->jacG2;
    DevBuf& aff = std::is_same<F, Fp>::value ? e->affG1 : e->affG2;
    int32_t rc; if ((rc = jac.reserve(n * sizeof(Jac<F>))) || (rc = aff.reserve(n * sizeof(Affine<F>)))) return rc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_synth_points<F>), dim3(nblk(n, 64)), dim3(64), 0, e->stream, g, start, (uint64_t)first, (uint64_t)stride, (uint32_t)n, jac.as<Jac<F>>());
    HIPCHK(hipGetLastError());
    if ((rc = e->normalize_dev<F>(jac.as<Jac<F>>(), n, aff.as<Affine<F>>()))) return rc;
    HIPCHK(hipMemcpyAsync(out, aff.p, n * sizeof(Affine<F>), hipMemcpyDeviceToHost, e->stream));
    return e->sync();
}
}  // extern "C++"
API int32_t ripp_synth_g1(uint64_t start, size_t first, size_t stride, size_t n, ripp_g1a* out) { return synth_points<Fp>(g1_generator(), start, first, stride, n, out); }
API int32_t ripp_synth_g2(uint64_t start, size_t first, size_t stride, size_t n, ripp_g2a* out) { return synth_points<Fp2>(g2_generator(), start, first, stride, n, out); }
API int32_t ripp_synth_fr(uint64_t seed, size_t first, size_t stride, size_t n, ripp_fr* out) {
    LOCK; ENGINE; if (n == 0) return RIPP_OK; if (!out) return RIPP_ERR_ARG;
    int32_t rc; if ((rc = e->tmpR.reserve(n * sizeof(Fr)))) return rc;
    hipLaunchKernelGGL(k_synth_fr, dim3(nblk(n, 256)), dim3(256), 0, e->stream, seed, (uint64_t)first, (uint64_t)stride, (uint32_t)n, e->tmpR.as<Fr>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, e->tmpR.p, n * sizeof(Fr), hipMemcpyDeviceToHost, e->stream));
    return e->sync();
}

}  // extern "C"
