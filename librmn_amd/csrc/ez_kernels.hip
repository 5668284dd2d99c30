/*
 * ez_kernels.hip -- hand-written HIP kernels (gfx950 / MI355X) for librmn's EZ interpolation
 * hot path, plus the thin C-ABI shim the C host front-end calls (ezhip_shim.h).
 *
 * Kernels:
 *   k_sepx<DEG,XR>  separable ("rectilinear on rectilinear") interpolation, the default: one thread block = one
 *                   256-column strip x several 16-row blocks; source rows staged once by LDS-DMA, x-pass results in
 *                   an fp64 LDS ring, y-pass with per-lane row records; all fields of a batch in one launch.
 *                   Replaces the do n=1,npts loops of ez_irgdint_3_w / ez_rgdint_3_w / ez_(i)rgdint_1_(n)w /
 *                   ez_rgdint_0 (reference src/interp, the .inc leaf kernels) when x depends on the target column
 *                   only and y on the target row only -- the BASELINE cfg1/2/4/5 shape.
 *   k_sep<DEG>      the same arithmetic as a (256 x 16) tile kernel with a rolling register window and a gather
 *                   path: fallback for plans k_sepx cannot take.
 *   k_pts           generic per-point interpolation at arbitrary (x,y): point-by-point restatement
 *                   of the 11 leaf kernels + zone handling (ez_defzones.c, ez_corrval*.c).
 *   k_locate        lat/lon -> source index space (ez_ll2rgd.inc, ez_ll2igd.inc, ez_cherche.inc,
 *                   ez_gfxyfll.c).
 *   k_wind_rotate   c_gdwdfuv + c_gduvfwd fused (ez_llwfgdw.inc, ez_gdwfllw.inc, ez_llwfgfw.c).
 *   k_polevals / k_minmax / k_fill   small reductions (ez_calcpoleval.inc, ez_aminmax.inc).
 *
 * No MFMA: there is no dense contraction on this path; the bound is HBM.
 * Wavefront = 64 lanes throughout.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "ezhip_shim.h"
#include "armn_dev.h"
#include "libm_exact.h"

/* Everything that restates reference arithmetic must not be contracted into FMAs; the separable
 * kernel uses explicit fma() where fusion is intended. */
#pragma clang fp contract(off)

/* ===================================================================================== */
/* runtime plumbing                                                                         */
/* ===================================================================================== */
static thread_local hipStream_t g_stream = nullptr;
static thread_local char g_err[256] = "";

/* every failed runtime call / launch of the library counts here: callers whose return value doubles as "not compressible" (armn_compress's -1) tell
 * an error from that answer by the count moving */
#include <atomic>
static std::atomic<unsigned> g_error_count{0};
extern "C" void ezhip_note_error(void) { g_error_count.fetch_add(1, std::memory_order_relaxed); }
extern "C" unsigned ezhip_error_count(void) { return g_error_count.load(std::memory_order_relaxed); }
static int set_err(hipError_t e, const char *what)
{
    if (e == hipSuccess) return 0;
    ezhip_note_error();
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -1;
}

extern "C" int ezhip_runtime_ok(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0;
}
extern "C" const char *ezhip_last_error(void) { return g_err; }
/* One process per GPU: grids' device mirrors, plans, located points and the per-thread workspaces are plain device pointers of the device that
 * was current when the library first touched the GPU.  A thread that comes in with ANOTHER current device would hand those pointers to kernels
 * of its own device; every compute entry point checks (need_device) and refuses loudly instead.  EZHIP_TEST_BOUND_DEVICE: tests only. */
#include <atomic>
static std::atomic<int> g_bound_dev{-1};
static void arm_thread_guard(void);
extern "C" int ezhip_bound_device_ok(const char *who)
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return -1; }
    int bound = g_bound_dev.load(std::memory_order_acquire);
    if (bound < 0) {
        const char *e = getenv("EZHIP_TEST_BOUND_DEVICE");
        int want = e ? atoi(e) : d, expected = -1;
        bound = g_bound_dev.compare_exchange_strong(expected, want, std::memory_order_acq_rel) ? want : expected;
    }
    if (bound == d) { arm_thread_guard(); return 0; }
    fprintf(stderr, "<%s> the library's grids, plans and workspaces of this process live on HIP device %d, but the calling thread's current device is %d: "
                    "one process per GPU (select the device before the first call and keep it)\n", who, bound, d);
    return -1;
}
extern "C" int ezhip_bound_device(void) { return g_bound_dev.load(std::memory_order_acquire); }

/* Per-thread state (workspaces, page-locked bounce buffers, the side stream, lists) is released when the host thread that owns it ends, and on request
 * (ezhip_thread_release): a service that runs calls on short-lived threads does not accumulate 2 x 16 MB of page-locked memory and its device workspaces
 * per thread that ever called.  Never from the MAIN thread and never in another process than the one that armed the guard: exit() runs the main
 * thread's thread_local destructors BEFORE the atexit handlers (so a flag set at exit is still clear there) with the HIP runtime on its way out, and a
 * fork()ed child inherits an armed guard over a runtime it cannot use.  The main thread's state goes with the process. */
#include <unistd.h>
#include <sys/syscall.h>
extern "C" void ezh_ez_thread_release(void);
extern "C" void ezhip_pack_release(void);
extern "C" void ezh_armn32_thread_release(void);
extern "C" void ezh_interpv_thread_release(void);
extern "C" void ezh_interpv_thread_release2(void);
static void kernels_thread_release(void);
static std::atomic<bool> g_process_exiting{false};
extern "C" void ezhip_thread_release(void)
{
    if (g_bound_dev.load(std::memory_order_acquire) < 0) return;                /* the library never touched the device */
    (void)hipDeviceSynchronize();
    ezh_ez_thread_release(); ezhip_pack_release(); ezh_armn32_thread_release(); ezh_interpv_thread_release(); ezh_interpv_thread_release2();
    kernels_thread_release();
}
namespace {
struct thread_guard {
    bool armed = false; pid_t pid = 0;
    ~thread_guard() {
        const pid_t me = getpid();
        if (!armed || g_process_exiting.load() || pid != me || (pid_t)syscall(SYS_gettid) == me) return;      /* main thread (tid == pid), or a forked child */
        ezhip_thread_release();
    }
};
thread_local thread_guard t_guard;
}
static void arm_thread_guard(void)
{
    static std::atomic<bool> once{false};
    if (!once.exchange(true)) atexit([]() { g_process_exiting.store(true); });
    t_guard.armed = true; t_guard.pid = getpid();
}

extern "C" void *ezhip_malloc(size_t nbytes)
{
    void *p = nullptr;
    if (set_err(hipMalloc(&p, nbytes ? nbytes : 4), "hipMalloc")) return nullptr;
    return p;
}
extern "C" void ezhip_free(void *d) { if (d) (void)hipFree(d); }
extern "C" int ezhip_h2d(void *d, const void *h, size_t n) { return set_err(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, g_stream), "h2d"); }
/* ---- results into the caller's ORDINARY (pageable) memory --------------------------------------------------------------------------
 * The device never writes the caller's ordinary memory: a copy of more than EZH_DIRECT_MAX bytes lands in two page-locked buffers of the calling thread
 * in turn (EZH_BOUNCE bytes each, truly asynchronous), and a small pool of host threads moves each chunk on into the caller's array while the next
 * chunk is in flight.  Why: copies into pageable memory make the runtime lock the caller's pages for the length of each copy, and with heaps that
 * allocate and free as a test session (or a long-running service) does, the copy engine's write now and then dies with "Memory access fault by GPU ...
 * Write access to a read-only page", the address inside the destination array -- 1 to 4 of 12 runs of the gpu test suite, any destination: a dense
 * 2.4 MB array, the rows of a pitched one, rows cut into ranges; writing each page from the CPU first did not stop it, page-locking the array for the
 * length of the call made it more frequent.  Arrays the caller registered (ezhip_register_host_buffer: locked once, for long) are written directly. */
#include <pthread.h>
#define EZH_DIRECT_MAX ((size_t)256 << 10)
#define EZH_BOUNCE_MAX ((size_t)32 << 20)
#define EZH_COPY_THREADS_MAX 16
static size_t EZH_BOUNCE = (size_t)16 << 20;      /* development: EZHIP_BOUNCE_MB, EZHIP_COPY_THREADS */
static int EZH_COPY_THREADS = 4;
extern "C" int ezh_host_is_pinned(const void *p, size_t n);
extern "C" void ezhip_touch_writable(void *h, size_t n)
{
    if (!h || !n) return;
    volatile char *q = (volatile char *)h;
    for (size_t off = 0; off < n; off += 4096) q[off] = q[off];
    q[n - 1] = q[n - 1];
}
namespace {
struct copy_pool {
    pthread_mutex_t use = PTHREAD_MUTEX_INITIALIZER;        /* one big copy at a time */
    pthread_mutex_t m = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t go = PTHREAD_COND_INITIALIZER, done = PTHREAD_COND_INITIALIZER;
    pthread_t th[EZH_COPY_THREADS_MAX];
    char *dst = nullptr; const char *src = nullptr; size_t n = 0;
    unsigned gen = 0; int pending = 0; bool started = false;
};
copy_pool g_pool;
/* the copy out of the page-locked buffer with non-temporal stores: the destination is written once and not read here, and an ordinary store first
 * fetches every line it is about to overwrite (EZHIP_COPY_PLAIN=1: memcpy) */
#include <immintrin.h>
__attribute__((target("avx2"))) void nt_copy_avx2(char *d, const char *s_, size_t n)
{
    size_t head = (32 - ((uintptr_t)d & 31)) & 31;
    if (head > n) head = n;
    if (head) { memcpy(d, s_, head); d += head; s_ += head; n -= head; }
    size_t k = 0;
    for (; k + 128 <= n; k += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(s_ + k)), b = _mm256_loadu_si256((const __m256i *)(s_ + k + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(s_ + k + 64)), e = _mm256_loadu_si256((const __m256i *)(s_ + k + 96));
        _mm256_stream_si256((__m256i *)(d + k), a); _mm256_stream_si256((__m256i *)(d + k + 32), b);
        _mm256_stream_si256((__m256i *)(d + k + 64), c); _mm256_stream_si256((__m256i *)(d + k + 96), e);
    }
    _mm_sfence();
    if (k < n) memcpy(d + k, s_ + k, n - k);
}
void copy_out(char *d, const char *s_, size_t n)
{
    static const int mode = (getenv("EZHIP_COPY_PLAIN") || !__builtin_cpu_supports("avx2")) ? 0 : 1;
    if (mode && n >= 4096) nt_copy_avx2(d, s_, n); else memcpy(d, s_, n);
}
void slice(int k, size_t n, size_t &o, size_t &l) { const size_t per = (n / EZH_COPY_THREADS + 63) & ~(size_t)63; o = per * k; l = o >= n ? 0 : (n - o < per || k == EZH_COPY_THREADS - 1 ? n - o : per); }
void *pool_worker(void *arg)
{
    const int k = (int)(intptr_t)arg;
    unsigned seen = 0;
    for (;;) {
        pthread_mutex_lock(&g_pool.m);
        while (g_pool.gen == seen) pthread_cond_wait(&g_pool.go, &g_pool.m);
        seen = g_pool.gen;
        char *d = g_pool.dst; const char *s_ = g_pool.src; const size_t n = g_pool.n;
        pthread_mutex_unlock(&g_pool.m);
        size_t o, l; slice(k, n, o, l);
        if (l) copy_out(d + o, s_ + o, l);
        pthread_mutex_lock(&g_pool.m);
        if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.done);
        pthread_mutex_unlock(&g_pool.m);
    }
    return nullptr;
}
void pool_memcpy(void *dst, const void *src, size_t n)
{
    if (n < ((size_t)1 << 20)) { memcpy(dst, src, n); return; }
    pthread_mutex_lock(&g_pool.use);
    if (!g_pool.started) {
        g_pool.started = true;
        for (int k = 1; k < EZH_COPY_THREADS; k++)
            if (pthread_create(&g_pool.th[k - 1], nullptr, pool_worker, (void *)(intptr_t)k) != 0) { g_pool.started = false; break; }
        if (g_pool.started) for (int k = 1; k < EZH_COPY_THREADS; k++) pthread_detach(g_pool.th[k - 1]);
    }
    if (!g_pool.started) { memcpy(dst, src, n); pthread_mutex_unlock(&g_pool.use); return; }
    pthread_mutex_lock(&g_pool.m);
    g_pool.dst = (char *)dst; g_pool.src = (const char *)src; g_pool.n = n; g_pool.pending = EZH_COPY_THREADS - 1; g_pool.gen++;
    pthread_cond_broadcast(&g_pool.go);
    pthread_mutex_unlock(&g_pool.m);
    size_t o, l; slice(0, n, o, l);
    if (l) copy_out((char *)dst + o, (const char *)src + o, l);
    pthread_mutex_lock(&g_pool.m);
    while (g_pool.pending) pthread_cond_wait(&g_pool.done, &g_pool.m);
    pthread_mutex_unlock(&g_pool.m);
    pthread_mutex_unlock(&g_pool.use);
}
/* pend: the LAST chunk of the previous copy, still in its buffer: it is moved to the caller's array while the next copy's first chunk is on its way (a call
 * that fetches a result in several row ranges keeps ONE pipeline going instead of draining it per range), at the latest by ezhip_sync */
thread_local struct { char *buf[2]; hipEvent_t ev[2]; bool ok; int next; struct { char *h; size_t len; int b; bool on; } pend; } t_bnc = {{nullptr, nullptr}, {nullptr, nullptr}, false, 0, {nullptr, 0, 0, false}};
bool bounce_ready()
{
    if (t_bnc.ok) return true;
    static bool tuned = false;
    if (!tuned) { tuned = true; const char *e = getenv("EZHIP_BOUNCE_MB"); if (e && atoi(e) >= 1 && atoi(e) <= 32) EZH_BOUNCE = (size_t)atoi(e) << 20;
        e = getenv("EZHIP_COPY_THREADS"); if (e && atoi(e) >= 1 && atoi(e) <= EZH_COPY_THREADS_MAX) EZH_COPY_THREADS = atoi(e); }
    for (int k = 0; k < 2; k++) {
        if (hipHostMalloc((void **)&t_bnc.buf[k], EZH_BOUNCE, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipEventCreateWithFlags(&t_bnc.ev[k], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    return t_bnc.ok = true;
}
}
static int bounce_finish()
{
    if (!t_bnc.pend.on) return 0;
    t_bnc.pend.on = false;
    if (set_err(hipEventSynchronize(t_bnc.ev[t_bnc.pend.b]), "d2h wait")) return -1;
    pool_memcpy(t_bnc.pend.h, t_bnc.buf[t_bnc.pend.b], t_bnc.pend.len);
    return 0;
}
extern "C" int ezhip_d2h(void *h, const void *d, size_t n)
{
    if (n <= EZH_DIRECT_MAX || ezh_host_is_pinned(h, n) || !bounce_ready())
        return set_err(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, g_stream), "d2h");
    for (size_t off = 0; off < n; off += EZH_BOUNCE) {
        const int b = t_bnc.next;
        t_bnc.next ^= 1;
        const size_t len = n - off < EZH_BOUNCE ? n - off : EZH_BOUNCE;
        if (t_bnc.pend.on && t_bnc.pend.b == b && bounce_finish()) return -1;        /* (cannot happen with two alternating buffers; kept for safety) */
        if (set_err(hipMemcpyAsync(t_bnc.buf[b], (const char *)d + off, len, hipMemcpyDeviceToHost, g_stream), "d2h") ||
            set_err(hipEventRecord(t_bnc.ev[b], g_stream), "d2h event")) return -1;
        if (bounce_finish()) return -1;                      /* the chunk before this one, of this copy or of the previous call, while this one is on its way */
        t_bnc.pend.h = (char *)h + off; t_bnc.pend.len = len; t_bnc.pend.b = b; t_bnc.pend.on = true;
    }
    /* the last chunk: moved out now, or -- row ranges fetched one after the other (EZHIP_HOST_UPLOADER) -- left pending for the next copy's first chunk to
     * overlap with; ezhip_sync drains it at the latest */
    static const bool keep = getenv("EZHIP_HOST_UPLOADER") != nullptr;
    return keep ? 0 : bounce_finish();
}
/* a blocking upload on a stream of the CALLING thread's own, for the host-pointer ABI's uploader thread (a hipMemcpy on the null stream would order itself
 * against every blocking stream of the process): binds the thread to the library's device, creates the stream on first use; ezhip_own_stream_release
 * before the thread ends */
static thread_local hipStream_t t_up = nullptr;
extern "C" int ezhip_h2d_blocking_own_stream(void *d, const void *h, size_t n)
{
    if (!t_up) {
        const int dev = g_bound_dev.load(std::memory_order_acquire);
        if (dev >= 0 && hipSetDevice(dev) != hipSuccess) return set_err(hipGetLastError(), "uploader thread: hipSetDevice");
        if (hipStreamCreateWithFlags(&t_up, hipStreamNonBlocking) != hipSuccess) { t_up = nullptr; return set_err(hipGetLastError(), "uploader thread: stream"); }
    }
    if (set_err(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, t_up), "h2d (uploader thread)")) return -1;
    return set_err(hipStreamSynchronize(t_up), "h2d (uploader thread)");
}
extern "C" void ezhip_own_stream_release(void) { if (t_up) { (void)hipStreamDestroy(t_up); t_up = nullptr; } }
extern "C" int ezhip_d2h_pinned(void *h, const void *d, size_t n) { return set_err(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, g_stream), "d2h"); }   /* h: page-locked memory of the library */
extern "C" int ezhip_d2d(void *dst, const void *src, size_t n) { return set_err(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, g_stream), "d2d"); }
extern "C" int ezhip_memset(void *d, int v, size_t n) { return set_err(hipMemsetAsync(d, v, n, g_stream), "memset"); }
extern "C" int ezhip_sync(void)
{
    const int bad = bounce_finish();                        /* a result chunk still in a bounce buffer reaches the caller's array here at the latest */
    return set_err(hipStreamSynchronize(g_stream), "sync") || bad ? -1 : 0;
}
extern "C" void ezhip_set_stream(void *s) { g_stream = (hipStream_t)s; }
extern "C" void *ezhip_get_stream(void) { return (void *)g_stream; }

/* A small kernel that only LATER work depends on (the polar wind rows: 44 us on two CUs) runs on a per-thread side
 * stream, forked from and joined back into the caller's stream with events:
 *   ezhip_side_begin()  side stream waits for everything queued on the caller's stream; launches now go to the side stream
 *   ezhip_side_end()    launches go to the caller's stream again
 *   ezhip_side_join()   the caller's stream waits for the side stream's work (no-op when nothing is pending) */
static thread_local hipStream_t t_side = nullptr, t_main_saved = nullptr;
static thread_local hipEvent_t t_ev_fork = nullptr, t_ev_join = nullptr;
static thread_local bool t_side_pending = false;
extern "C" int ezhip_side_begin(void)
{
    if (!t_side) {
        if (hipStreamCreateWithFlags(&t_side, hipStreamNonBlocking) != hipSuccess) return set_err(hipGetLastError(), "side stream");
        if (hipEventCreateWithFlags(&t_ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&t_ev_join, hipEventDisableTiming) != hipSuccess)
            return set_err(hipGetLastError(), "side events");
    }
    if (hipEventRecord(t_ev_fork, g_stream) != hipSuccess || hipStreamWaitEvent(t_side, t_ev_fork, 0) != hipSuccess) return set_err(hipGetLastError(), "fork");
    t_main_saved = g_stream; g_stream = t_side;
    return 0;
}
extern "C" int ezhip_side_end(void)
{
    hipError_t e = hipEventRecord(t_ev_join, t_side);
    g_stream = t_main_saved;
    t_side_pending = true;
    return e == hipSuccess ? 0 : set_err(e, "side end");
}
extern "C" int ezhip_side_join(void)
{
    if (!t_side_pending) return 0;
    t_side_pending = false;
    return set_err(hipStreamWaitEvent(g_stream, t_ev_join, 0), "join");
}
extern "C" void *ezhip_host_alloc(size_t n)
{
    void *p = nullptr;
    if (set_err(hipHostMalloc(&p, n ? n : 4, hipHostMallocDefault), "hipHostMalloc")) return nullptr;
    return p;
}
extern "C" void ezhip_host_free(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int ezhip_host_pin(void *p, size_t n) { ezhip_touch_writable(p, n); return set_err(hipHostRegister(p, n, hipHostRegisterDefault), "hipHostRegister"); }
extern "C" int ezhip_host_unpin(void *p) { return set_err(hipHostUnregister(p), "hipHostUnregister"); }

#define LAUNCH_CHECK(what) set_err(hipGetLastError(), what)

/* Device-side error word: one int in pinned host memory that a kernel sets (system-scope store) when it has to give up --
 * today only k_sepx's bounded wait for the in-launch pole sums.  Sticky: ezhip_device_error() reports and clears it; the
 * entry points check it at entry and, where they synchronise anyway, before they return. */
static int *g_dev_err = nullptr;
static int *dev_err_word(void)
{
    static int *volatile word = nullptr;
    if (!word) {
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        memset(p, 0, 64);
        int *expect = nullptr;
        if (!__atomic_compare_exchange_n((int **)&word, &expect, (int *)p, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) (void)hipHostFree(p);
    }
    g_dev_err = word;
    return word;
}
extern "C" int ezhip_device_error(void)
{
    int *w = dev_err_word();
    if (!w) return 0;
    return __atomic_exchange_n(w, 0, __ATOMIC_ACQ_REL);
}

/* ===================================================================================== */
/* block reduction helpers (wave = 64)                                                      */
/* ===================================================================================== */
__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

/* sum over a 256-thread block; result valid in every thread */
__device__ double block_sum_256(double v, double *lds4)
{
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds4[w] = v;
    __syncthreads();
    return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

/* ez_calcpoleval (src/interp/ez_calcpoleval.inc:21-48).  The reference accumulates the row
 * SEQUENTIALLY in REAL; any tree order changes the low bits (measured 2.9e-6 relative at ni = 4400),
 * so the sum is kept sequential: the block stages the row through LDS in 1024-element chunks (the
 * Z-on-E variant stages the REAL products z(i)*(ax(i+1)-ax(i))) and lane 0 adds them in index order.
 * ~4 cycles per dependent v_add_f32: ~8 us for ni = 4400, hidden because the special blocks that
 * need it are dispatched first and run beside the main blocks.  Result valid in every thread. */
#define POLE_CHUNK 8192          /* floats staged per pass by k_polevals: a whole source row up to ni = 8192 (one load latency) */
/* lds: 16-byte aligned, chunk + 4 floats; chunk a multiple of 4 */
__device__ float block_poleval(const float *zrow, int ni, int weighted, const float *ax, float *lds, const int chunk)
{
    const int n = weighted ? ni - 1 : ni;
    float s = 0.0f;
    for (int base = 0; base < n; base += chunk) {
        const int m = min(chunk, n - base);
        __syncthreads();
        /* eight elements in flight per lane (as a plain copy loop the compiler waits for each load before it issues the next: one memory
         * round trip per blockDim elements, 14 - 17 of them for a 4400-point row, in front of the sequential sum every special row waits for) */
        for (int k0 = threadIdx.x; k0 < m; k0 += 8 * blockDim.x) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = base + min(k0 + u * (int)blockDim.x, m - 1);
                x[u] = weighted ? zrow[i] * (ax[i + 1] - ax[i]) : zrow[i];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int k = k0 + u * (int)blockDim.x; if (k < m) lds[k] = x[u]; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            /* the reference's sum is a sequential REAL accumulation: one lane, 64 elements (16 x ds_read_b128) in
             * flight per dependent-add burst; the per-element LDS round trip of the scalar loop cost 20 us per row */
            const float4 *l4 = (const float4 *)lds;
            const int m4 = m >> 2;
            int q = 0;
            for (; q + 16 <= m4; q += 16) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = l4[q + u];
#pragma unroll
                for (int u = 0; u < 16; u++) { s = s + v[u].x; s = s + v[u].y; s = s + v[u].z; s = s + v[u].w; }
            }
            for (int k = 4 * q; k < m; k++) s = s + lds[k];
        }
    }
    if (threadIdx.x == 0) {
        if (weighted) { float span = ax[ni - 1] - ax[0]; if (span != 0.0f) s = s / span; }
        else s = s / (1.0f * (float)ni);
        lds[chunk] = s;
    }
    __syncthreads();
    s = lds[chunk];
    __syncthreads();
    return s;
}

/* two rows at once (the pole values of the two components of a wind): the same sequential REAL sums, row a by lane 0 of wave 0 and row b by lane 0 of wave 1 at the
 * same time (each row has its half of the staging buffer: lds holds 2 * chunk + 4 floats): the two dependent-add chains of ~ni terms run side by side */
__device__ void block_poleval2(const float *rowa, const float *rowb, int ni, int weighted, const float *ax, float *lds, const int chunk, float &ra, float &rb, const int nthr /* threads taking part (the first nthr of the block) */)
{
    const int n = weighted ? ni - 1 : ni;
    float s = 0.0f;
    for (int base = 0; base < n; base += chunk) {
        const int m = min(chunk, n - base);
        __syncthreads();
        for (int k0 = threadIdx.x; k0 < m; k0 += 4 * nthr) {
            float x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + min(k0 + u * nthr, m - 1);
                const float w = weighted ? ax[i + 1] - ax[i] : 1.0f;
                x[u] = weighted ? rowa[i] * w : rowa[i]; y[u] = weighted ? rowb[i] * w : rowb[i];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int k = k0 + u * nthr; if (k < m) { lds[k] = x[u]; lds[chunk + k] = y[u]; } }
        }
        __syncthreads();
        if (threadIdx.x == 0 || threadIdx.x == 64) {
            const float *mine = lds + (threadIdx.x ? chunk : 0);
            const float4 *l4 = (const float4 *)mine;
            const int m4 = m >> 2;
            int q = 0;
            for (; q + 16 <= m4; q += 16) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = l4[q + u];
#pragma unroll
                for (int u = 0; u < 16; u++) { s = s + v[u].x; s = s + v[u].y; s = s + v[u].z; s = s + v[u].w; }
            }
            for (int k = 4 * q; k < m; k++) s = s + mine[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 || threadIdx.x == 64) {
        if (weighted) { float span = ax[ni - 1] - ax[0]; if (span != 0.0f) s = s / span; }
        else s = s / (1.0f * (float)ni);
        lds[2 * chunk + (threadIdx.x ? 1 : 0)] = s;
    }
    __syncthreads();
    ra = lds[2 * chunk]; rb = lds[2 * chunk + 1];
    __syncthreads();
}

/* order-preserving float <-> uint32 keys (min / max by integer compare) */
__device__ __forceinline__ unsigned f2key(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

/* ===================================================================================== */
/* k_sep : separable interpolation                                                          */
/* ===================================================================================== */
#define SEP_BLOCK 256

struct ColTaps { int i0, i1, i2, i3; double w0, w1, w2, w3; };

__device__ __forceinline__ ColTaps load_col(const int *cidx, const double *cw, int ni_dst, int c)
{
    ColTaps t;
    t.i0 = cidx[c]; t.i1 = cidx[ni_dst + c]; t.i2 = cidx[2 * ni_dst + c]; t.i3 = cidx[3 * ni_dst + c];
    t.w0 = cw[c]; t.w1 = cw[ni_dst + c]; t.w2 = cw[2 * ni_dst + c]; t.w3 = cw[3 * ni_dst + c];
    return t;
}

/* x-direction pass on one source row (global memory or an LDS patch row; indices are relative to zrow) */
template <int DEG, class T>
__device__ __forceinline__ double xpass(const T *__restrict__ zrow, const ColTaps &t)
{
    if (DEG == 0) return (double)zrow[t.i0];
    if (DEG == 1) {   /* zlin8.cdk: z1 + (z2 - z1) * dx, exactly as the reference evaluates it */
        double z1 = (double)zrow[t.i0], z2 = (double)zrow[t.i1];
        return z1 + (z2 - z1) * t.w0;
    }
    double z0 = (double)zrow[t.i0], z1 = (double)zrow[t.i1], z2 = (double)zrow[t.i2], z3 = (double)zrow[t.i3];
    return fma(t.w3, z3, fma(t.w2, z2, fma(t.w1, z1, t.w0 * z0)));
}

/* per-row metadata of one row-block, staged in LDS once (a per-row chain of dependent global /
 * scalar loads measured ~1000 cycles per row and made the first version latency-bound) */
struct RowInfo { int rbase[EZHIP_SEP_ROWS]; int rflag[EZHIP_SEP_ROWS]; double rw[4][EZHIP_SEP_ROWS]; };

/* The main-row loop: rolling window of x-pass results over the source rows; `rowptr(s)` yields the
 * base pointer of source row s (global memory, or the block's LDS patch). */
template <int DEG, class RowPtr>
__device__ __forceinline__ void sep_rows(const ezhip_sep_plan &p, const RowInfo &ri, const ColTaps &t, RowPtr rowptr, int r0, int r1,
                                         float *__restrict__ zout, int c, bool cvalid, bool cdehors, float fillv)
{
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    int cur = -(1 << 28);
    for (int k = 0; k < r1 - r0; k++) {
        if (ri.rflag[k]) continue;                      /* uniform: row handled as special */
        const int r = r0 + k;
        const int jb = ri.rbase[k];
        double val;
        if (DEG == 0) {
            val = xpass<0>(rowptr(jb), t);
        } else if (DEG == 1) {
            int d = jb - cur;
            if (d != 0) {
                if (d == 1) { t0 = t1; t1 = xpass<1>(rowptr(jb + 1), t); }
                else { t0 = xpass<1>(rowptr(jb), t); t1 = xpass<1>(rowptr(jb + 1), t); }
                cur = jb;
            }
            val = t0 + (t1 - t0) * ri.rw[0][k];
        } else {
            int d = jb - cur;
            if (d != 0) {
                if (d == 1) { t0 = t1; t1 = t2; t2 = t3; t3 = xpass<3>(rowptr(jb + 3), t); }
                else if (d == 2) { t0 = t2; t1 = t3; t2 = xpass<3>(rowptr(jb + 2), t); t3 = xpass<3>(rowptr(jb + 3), t); }
                else if (d == 3) { t0 = t3; t1 = xpass<3>(rowptr(jb + 1), t); t2 = xpass<3>(rowptr(jb + 2), t); t3 = xpass<3>(rowptr(jb + 3), t); }
                else { t0 = xpass<3>(rowptr(jb), t); t1 = xpass<3>(rowptr(jb + 1), t); t2 = xpass<3>(rowptr(jb + 2), t); t3 = xpass<3>(rowptr(jb + 3), t); }
                cur = jb;
            }
            val = fma(ri.rw[3][k], t3, fma(ri.rw[2][k], t2, fma(ri.rw[1][k], t1, ri.rw[0][k] * t0)));
        }
        if (cvalid) zout[(size_t)r * p.ni_dst + c] = cdehors ? fillv : (float)val;
    }
}

/* x-direction pass on one LDS patch row with four CONSECUTIVE taps */
template <int DEG>
__device__ __forceinline__ double xrow(const float *pr, const double (&cw)[4])
{
    if (DEG == 0) return (double)pr[0];
    if (DEG == 1) { double z1 = (double)pr[0], z2 = (double)pr[1]; return z1 + (z2 - z1) * cw[0]; }
    return fma(cw[3], (double)pr[3], fma(cw[2], (double)pr[2], fma(cw[1], (double)pr[1], cw[0] * (double)pr[0])));
}

/* Staged fast path of one (column-block, row-block): the source patch is in LDS and the four taps of
 * every column are consecutive patch columns.  Lean by construction -- three earlier versions were
 * bound by (1) dependent global loads per row, (2) ~46 VALU per row, (3) ~40 serialized scalar-cache
 * round trips per wave (13 us wave lifetime for 145 VALU instructions):
 *   - lane k (< 16) of every wave loads the metadata {jb, flag, w[4]} of row k ONCE (one coalesced
 *     vector load); each statically unrolled row then broadcasts it with v_readlane into SGPRs:
 *     no memory instruction and no wait in the row loop, control flow on jb is scalar, the weights
 *     feed v_fma_f64 as scalar operands;
 *   - one VGPR address per new source row, taps at immediate offsets. */
__device__ __forceinline__ double readlane_f64(double v, int k)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

/* LDS-DMA of one dword per lane, issued through inline asm ON PURPOSE: for the builtin the compiler tracks the
 * LDS destination and conservatively inserts `s_waitcnt vmcnt(0)` in front of later ds_reads that may alias it
 * (it cannot tell the two patch buffers apart), which serialises the software pipeline and drains the stores.
 * Completion is ordered by the hand-placed vmcnt waits + the workgroup barrier instead.
 * LDS address of lane l = lds_byte_addr + 4 l ; global address = base + voff_bytes(l).  m0 is written without a
 * clobber (it is a reserved register for inline asm); nothing else in these kernels uses m0. */
__device__ __forceinline__ void lds_dma_dword(const float *base_uniform, unsigned voff_bytes, unsigned lds_byte_addr)
{
    const char *addr = (const char *)base_uniform + voff_bytes;        /* per-lane 64-bit address */
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                 :: "v"(addr), "s"(__builtin_amdgcn_readfirstlane((int)lds_byte_addr)) : "memory");
}
/* the 16-byte form (gfx950): lane l brings 16 bytes to lds_byte_addr + 16 l */
__device__ __forceinline__ void lds_dma_dwordx4(const float *base_uniform, unsigned voff_bytes, unsigned lds_byte_addr)
{
    const char *addr = (const char *)base_uniform + voff_bytes;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(addr), "s"(__builtin_amdgcn_readfirstlane((int)lds_byte_addr)) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }

#define SEP_QCH ((EZHIP_SEP_WMAX + 63) / 64)   /* 64-column chunks per staged row */

/* 16 statically unrolled target rows of one row-block from the LDS patch (taps = consecutive patch
 * columns).  Row metadata sits in lane k of `ri` and is broadcast with v_readlane: no memory
 * instruction or wait inside the row loop, scalar control flow, scalar weight operands. */
template <int DEG>
__device__ __forceinline__ void sep_rows_staged(const ezhip_sep_plan &p, const ezhip_rowinfo &mine, const float *patch, int wstride,
                                                int off0, const double (&cw)[4], int r0,
                                                float *__restrict__ zout, int c, bool cvalid, bool cdehors, float fillv)
{
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    int cur = -(1 << 28);
    const float *pcol = patch + off0;
    float *orow = zout + (size_t)r0 * p.ni_dst + c;
#define XP(row) xrow<DEG>(pcol + (row) * wstride, cw)
#pragma unroll
    for (int k = 0; k < EZHIP_SEP_ROWS; k++, orow += p.ni_dst) {
        if (__builtin_amdgcn_readlane(mine.flag, k)) continue;   /* scalar: special row, or past the last row */
        const int jb = __builtin_amdgcn_readlane(mine.jb, k);     /* patch-relative first source row */
        double val;
        if (DEG == 0) {
            val = XP(jb);
        } else if (DEG == 1) {
            const int d = jb - cur;
            if (d != 0) {
                if (d == 1) { t0 = t1; t1 = XP(jb + 1); }
                else { t0 = XP(jb); t1 = XP(jb + 1); }
                cur = jb;
            }
            val = t0 + (t1 - t0) * readlane_f64(mine.w[0], k);
        } else {
            const int d = jb - cur;
            if (d != 0) {
                if (d == 1) { t0 = t1; t1 = t2; t2 = t3; t3 = XP(jb + 3); }
                else if (d == 2) { t0 = t2; t1 = t3; t2 = XP(jb + 2); t3 = XP(jb + 3); }
                else if (d == 3) { t0 = t3; t1 = XP(jb + 1); t2 = XP(jb + 2); t3 = XP(jb + 3); }
                else { t0 = XP(jb); t1 = XP(jb + 1); t2 = XP(jb + 2); t3 = XP(jb + 3); }
                cur = jb;
            }
            val = fma(readlane_f64(mine.w[3], k), t3, fma(readlane_f64(mine.w[2], k), t2,
                  fma(readlane_f64(mine.w[1], k), t1, readlane_f64(mine.w[0], k) * t0)));
        }
        if (cvalid) *orow = cdehors ? fillv : (float)val;
    }
#undef XP
}

/* Straight-line variant of sep_rows_staged for the common row-block: no flagged row and every row advances
 * the source window by 0 or 1 (any up-sampling target).  The window shift is a SELECT on a scalar condition
 * and the newest source row is x-passed for every target row (16 x-passes instead of ~14), so the 16 unrolled
 * rows contain no branch: the compiler hoists all LDS reads and overlaps the rows (the branchy version chains
 * one LDS round trip per row; measured ~700-1000 cycles per row). */
template <int DEG>
__device__ __forceinline__ void sep_rows_simple(const ezhip_sep_plan &p, const ezhip_rowinfo &mine, const float *patch, int wstride,
                                                int off0, const double (&cw)[4], int r0,
                                                float *__restrict__ zout, int c, bool cvalid, bool cdehors, float fillv)
{
    const float *pcol = patch + off0;
    float *orow = zout + (size_t)r0 * p.ni_dst + c;
    const int jb0 = __builtin_amdgcn_readlane(mine.jb, 0);
#define XP(row) xrow<DEG>(pcol + (row) * wstride, cw)
    double t0, t1, t2, t3;
    if (DEG == 3) { t0 = XP(jb0); t1 = XP(jb0 + 1); t2 = XP(jb0 + 2); t3 = XP(jb0 + 3); }
    else { t0 = XP(jb0); t1 = XP(jb0 + 1); t2 = t3 = 0.0; }
    int prev = jb0;
#pragma unroll
    for (int k = 0; k < EZHIP_SEP_ROWS; k++, orow += p.ni_dst) {
        const int jb = __builtin_amdgcn_readlane(mine.jb, k);
        const bool adv = jb != prev;                       /* scalar; jb - prev is 0 or 1 by construction */
        prev = jb;
        double val;
        if (DEG == 1) {
            const double tn = XP(jb + 1);
            t0 = adv ? t1 : t0; t1 = adv ? tn : t1;
            val = t0 + (t1 - t0) * readlane_f64(mine.w[0], k);
        } else {
            const double tn = XP(jb + 3);
            t0 = adv ? t1 : t0; t1 = adv ? t2 : t1; t2 = adv ? t3 : t2; t3 = adv ? tn : t3;
            val = fma(readlane_f64(mine.w[3], k), t3, fma(readlane_f64(mine.w[2], k), t2,
                  fma(readlane_f64(mine.w[1], k), t1, readlane_f64(mine.w[0], k) * t0)));
        }
        if (cvalid) *orow = cdehors ? fillv : (float)val;
    }
#undef XP
}

/* gather fallback for one row-block (patch not usable: strong down-sampling, non-consecutive literal
 * seam taps, or more source rows than the patch holds) */
template <int DEG>
__device__ void sep_rowblock_gather(const ezhip_sep_plan &p, RowInfo &ri, float *__restrict__ zout, const float *__restrict__ zin,
                                    int by, int cc, int c, bool cvalid, bool cdehors, float fillv)
{
    const int r0 = by * EZHIP_SEP_ROWS, r1 = min(r0 + EZHIP_SEP_ROWS, p.nj_dst);
    __syncthreads();
    if (threadIdx.x < EZHIP_SEP_ROWS) {
        const int k = threadIdx.x, r = r0 + k;
        const bool ok = r < r1;
        ri.rflag[k] = ok ? p.rflag[r] : 1;
        ri.rbase[k] = ok ? p.rbase[r] : 0;
        for (int q = 0; q < 4; q++) ri.rw[q][k] = ok ? p.rw[q * p.nj_dst + r] : 0.0;
    }
    const ColTaps t = load_col(p.cidx, p.cw, p.ni_dst, cc);
    __syncthreads();
    const int nis = p.ni_src;
    sep_rows<DEG>(p, ri, t, [&](int s) { return zin + (size_t)s * nis; }, r0, r1, zout, c, cvalid, cdehors, fillv);
}

/* token output of k_sepx (out_mode 3): compact_float's quantisation parameters of the field, its token array */
struct QuantP { double minF, mul; unsigned short *tok16; unsigned colbase; };
__device__ __forceinline__ unsigned quant16(float v, const QuantP &q)
{   /* compact.tmplc:285-300: (int64)((double(a) - min) * mulFactor); a >= min, so the unsigned conversion truncates identically */
    return __double2uint_rz(((double)v - q.minF) * q.mul) & 0xFFFFu;
}

/* special target rows of one column block: polar strips, pole rows, fully-outside rows (`ispecial` indexes
 * p.special).  Pole values come precomputed (p.polevals). */
template <int DEG, int OUT = 0>
__device__ float sep_special(const ezhip_sep_plan &p, float *__restrict__ zout, const float *__restrict__ zin, int ispecial,
                             int c, int cc, bool cvalid, float fillv, const QuantP &qp = QuantP())
{
    const int nis = p.ni_src;
    /* ---- special rows: polar strips, pole rows, fully-outside rows ------------------------- */
    const ezhip_special_row sr = p.special[ispecial];
    float outv;
    if (sr.kind == 3) {
        outv = fillv;
    } else {
        bool need_n = (sr.kind == 1), need_s = (sr.kind == 2);
        if (sr.kind == 0)
            for (int k = 0; k < 4; k++) { need_n |= (sr.tap[k] == EZ_ROW_POLE_N); need_s |= (sr.tap[k] == EZ_ROW_POLE_S); }
        float pole_n = 0.f, pole_s = 0.f;
        if (!p.vector_mode) {       /* pole values: precomputed once per field by k_polevals (a sequential REAL sum: ~10 us) */
            if (need_n) pole_n = p.pole_timeout ? __builtin_nanf("") : (p.pole_inline ? p.pole_now[0] : p.polevals[0]);
            if (need_s) pole_s = p.pole_timeout ? __builtin_nanf("") : (p.pole_inline ? p.pole_now[1] : p.polevals[1]);
        }
        if (sr.kind == 1) outv = pole_n;
        else if (sr.kind == 2) outv = pole_s;
        else {
            const ColTaps t = load_col(p.cidx_s, p.cw_s, p.ni_dst, cc);
            double tv[4];
            const int ntap = (DEG == 3) ? 4 : (DEG == 1 ? 2 : 1);
            for (int k = 0; k < ntap; k++) {
                int row = sr.tap[k];
                if (row >= 0) tv[k] = xpass<DEG>(zin + (size_t)row * nis, t);
                else if (p.vector_mode) tv[k] = xpass<DEG>(row == EZ_ROW_POLE_N ? p.pole_row_n : p.pole_row_s, t);
                else tv[k] = (double)(row == EZ_ROW_POLE_N ? pole_n : pole_s);   /* constant row interpolates to itself */
            }
            double val;
            if (DEG == 0) val = tv[0];
            else if (DEG == 1) val = tv[0] + (tv[1] - tv[0]) * sr.w[0];
            else val = fma(sr.w[3], tv[3], fma(sr.w[2], tv[2], fma(sr.w[1], tv[1], sr.w[0] * tv[0])));
            outv = (float)val;
        }
    }
    if (cvalid) {
        if (OUT == 3) qp.tok16[((unsigned)sr.row * (unsigned)p.ni_dst + (unsigned)c) ^ 1u] = (unsigned short)quant16(outv, qp);
        else if (OUT != 2) zout[(size_t)sr.row * p.ni_dst + c] = outv;
    }
    return outv;       /* lanes past the last column carry the value of the last column */
}

/* k_sep: the tile kernel, one (256-column block, 16-row block) per thread block: stage the source patch, barrier,
 * compute + store.  It is the FALLBACK of k_sepx (below) for plans that one cannot take -- column blocks whose taps
 * are not consecutive (literal seam remaps: gathers), source windows taller than the patch (strong down-sampling) --
 * and was the round-1 default before k_sepx (53.5 us per cfg2 bicubic field against 35). */
template <int DEG>
__global__ __launch_bounds__(SEP_BLOCK) void k_sep(ezhip_sep_plan p, float *__restrict__ zout,
                                                   const float *__restrict__ zin)
{
    extern __shared__ float smem[];        /* the patch: p.patch_elems floats */
    __shared__ RowInfo ri_lds;             /* gather fallback only */
    const int c = blockIdx.x * SEP_BLOCK + threadIdx.x;
    const bool cvalid = c < p.ni_dst;
    const int cc = cvalid ? c : p.ni_dst - 1;
    const int nis = p.ni_src;
    const float fillv = p.fill ? *p.fill : 0.0f;

    /* special rows occupy the FIRST blockIdx.y values */
    if ((int)blockIdx.y >= p.n_special) {
        const int by = blockIdx.y - p.n_special;
        const bool cdehors = p.cflag[cc] != 0;
        const int base = p.blk_base[blockIdx.x], W = p.blk_w[blockIdx.x];
        const int s0 = p.brow_s0[by], n = p.brow_n[by];
        if (base < 0 || n <= 0) { sep_rowblock_gather<DEG>(p, ri_lds, zout, zin, by, cc, c, cvalid, cdehors, fillv); return; }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        unsigned coloff[SEP_QCH];
#pragma unroll
        for (int q = 0; q < SEP_QCH; q++) {          /* source column of patch column lane + 64 q (seam unrolled) */
            int col = base + lane + 64 * q;
            if (col >= nis) col -= nis;
            coloff[q] = (unsigned)col;
        }
        const int off0 = p.coff[cc];
        const double cw[4] = {p.cw[cc], p.cw[p.ni_dst + cc], p.cw[2 * p.ni_dst + cc], p.cw[3 * p.ni_dst + cc]};
        /* all loads issued up-front: no vector load is issued after a store, so no wave waits on a store acknowledgement */
        for (int row = wv; row < n; row += SEP_BLOCK / 64) {
            const float *zr = zin + (size_t)(s0 + row) * nis;      /* uniform base */
            float *prow = smem + row * p.wstride + lane;
#pragma unroll
            for (int q = 0; q < SEP_QCH; q++)
                if (lane + 64 * q < W) prow[64 * q] = zr[coloff[q]];
        }
        const ezhip_rowinfo mine = p.rowinfo[(size_t)by * EZHIP_SEP_ROWS + (threadIdx.x & (EZHIP_SEP_ROWS - 1))];
        __syncthreads();
        if (DEG != 0 && mine.pad0)      /* pad0 (uniform across lanes): row-block qualifies for the straight-line body */
            sep_rows_simple<DEG>(p, mine, smem, p.wstride, off0, cw, by * EZHIP_SEP_ROWS, zout, c, cvalid, cdehors, fillv);
        else
            sep_rows_staged<DEG>(p, mine, smem, p.wstride, off0, cw, by * EZHIP_SEP_ROWS, zout, c, cvalid, cdehors, fillv);
        return;
    }
    (void)sep_special<DEG>(p, zout, zin, blockIdx.y, c, cc, cvalid, fillv);
}

/* ===================================================================================== */
/* k_sepx : separable interpolation, x-pass results in an LDS ring, y-pass with per-lane rows */
/* ===================================================================================== */
/* Why (measured on MI355X, profiles/r01_*): k_sep keeps a rolling 4-row window of x-pass results in registers;
 * shifting it costs 8 v_cndmask per target row, the row-uniform y weights 8 v_readlane (or one scalar load and one
 * lgkmcnt(0) per row), the newest source row is x-passed once per TARGET row and every row-block re-stages its halo
 * rows: ~33 VALU + ~25 SALU instructions per point, 4 waves per SIMD -> issue-bound, 26 us of VALU next to 34 us of
 * memory time per cfg2 field.  Here
 *   - a thread block owns a 256-column strip x x_rb consecutive row-blocks; each source row of the strip is staged
 *     (LDS-DMA, one patch buffer) and x-interpolated exactly once per thread block;
 *   - x-pass: thread = target column (x weights are per-thread constants); the fp64 result of source row s goes to
 *     the ring T[s mod x_tr][column] in LDS;
 *   - y-pass: lane = (target row, column): a wave covers 2 target rows x 32 columns per instruction, so the y weights,
 *     the four ring-tap addresses and the store offset of a row are PER-LANE registers, loaded once per row-block
 *     from a 64-byte row record that rides the DMA; the 8 column groups of a row pair are immediate offsets.  Per 64
 *     points: 4 ds_read_b64 + 4 fp64 ops + cvt + store, no scalar work at all;
 *   - per row-block: wait DMA(i) -> barrier -> x-pass of the NEW rows -> barrier -> issue DMA(i+1) -> y-pass + 16
 *     stores per wave.  vmcnt is in-order: `vmcnt(16)` = "everything but my 16 youngest stores", so the DMA wait never
 *     waits for a store acknowledgement.  Rows that are not main rows (special rows, padding) re-store a neighbouring
 *     main row (the record names the row to write).
 * The arithmetic is the one of k_sep (same fma chains), so results are bit-identical to it. */
#define CONSTP(T, ptr) ((const __attribute__((address_space(4))) T *)(ptr))

/* y-pass of one wave and one row-block: 2 row pairs x 8 column groups of 32.  `myrec` = this lane's row record of
 * the first pair (the second pair is 8 records further), `tcol` = ring base of the lane's column in group 0.
 * SLOW adds what few blocks need: the DEHORS fill select, the column bound of the last strip, the debug knock-out. */
/* OUT (what the launch leaves behind, ezhip_sep_plan.out_mode): 0 floats, 1 floats + min/max partials, 2 min/max partials
 * only (nothing stored), 3 compact_float's 16-bit tokens */
template <int DEG, int XR, bool SLOW, int OUT>
__device__ __forceinline__ void sepx_ypass(const float *myrec, const double *tcol, float *zcol, float fillv, unsigned dmask,
                                           int l32, int ncol_valid, bool nostore, float &vmin, float &vmax, const QuantP &qp)
{
    constexpr bool STATS = OUT == 1 || OUT == 2;
#pragma unroll
    for (int h = 0; h < XR / 8; h++) {
        const float4 *r4 = (const float4 *)(myrec + h * 8 * 16);
        const float4 ra = r4[0], rb = r4[1], rc = r4[2];      /* w[0..3] | tap byte offsets */
        const unsigned o_off = (unsigned)__float_as_int(myrec[h * 8 * 16 + 12]);
        const double w0 = __hiloint2double(__float_as_int(ra.y), __float_as_int(ra.x));
        const double w1 = __hiloint2double(__float_as_int(ra.w), __float_as_int(ra.z));
        const double w2 = __hiloint2double(__float_as_int(rb.y), __float_as_int(rb.x));
        const double w3 = __hiloint2double(__float_as_int(rb.w), __float_as_int(rb.z));
        const char *tb = (const char *)tcol;
        const double *tp0 = (const double *)(tb + __float_as_int(rc.x)), *tp1 = (const double *)(tb + __float_as_int(rc.y));
        const double *tp2 = (const double *)(tb + __float_as_int(rc.z)), *tp3 = (const double *)(tb + __float_as_int(rc.w));
        float *orow = zcol + o_off;                /* element offset of the target row */
        double t[8][4];
        if (SLOW) {
#pragma unroll
            for (int g = 0; g < 8; g++) {
                t[g][0] = tp0[32 * g];
                if (DEG >= 1) t[g][1] = tp1[32 * g];
                if (DEG == 3) { t[g][2] = tp2[32 * g]; t[g][3] = tp3[32 * g]; }
            }
        } else {
            /* the ring taps as 32 (16) single ds_read_b64 through inline asm: the compiler pairs them into ds_read2_b64,
             * which costs 8.1 LDS cycles per wave on gfx950 against 2 x 2.2 for two ds_read_b64 (tools/irate).  One wait
             * for the row pair; LDS reads of the other waves cover it. */
            const unsigned a0 = lds_addr_of(tp0), a1 = lds_addr_of(tp1), a2 = lds_addr_of(tp2), a3 = lds_addr_of(tp3);
#define RD8(A, J) asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:256\n\tds_read_b64 %2, %8 offset:512\n\tds_read_b64 %3, %8 offset:768\n\t" \
                               "ds_read_b64 %4, %8 offset:1024\n\tds_read_b64 %5, %8 offset:1280\n\tds_read_b64 %6, %8 offset:1536\n\tds_read_b64 %7, %8 offset:1792" \
                               : "=&v"(t[0][J]), "=&v"(t[1][J]), "=&v"(t[2][J]), "=&v"(t[3][J]), "=&v"(t[4][J]), "=&v"(t[5][J]), "=&v"(t[6][J]), "=&v"(t[7][J]) \
                               : "v"(A) : "memory")
            RD8(a0, 0);
            if (DEG >= 1) RD8(a1, 1);
            if (DEG == 3) { RD8(a2, 2); RD8(a3, 3); }
#undef RD8
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            /* tie the values to the wait so that no consumer is scheduled above it */
#pragma unroll
            for (int g = 0; g < 8; g++) {
                if (DEG == 3) asm volatile("" : "+v"(t[g][0]), "+v"(t[g][1]), "+v"(t[g][2]), "+v"(t[g][3]));
                else if (DEG == 1) asm volatile("" : "+v"(t[g][0]), "+v"(t[g][1]));
                else asm volatile("" : "+v"(t[g][0]));
            }
        }
#pragma unroll
        for (int g = 0; g < 8; g++) {
            double val;
            if (DEG == 0) val = t[g][0];
            else if (DEG == 1) val = t[g][0] + (t[g][1] - t[g][0]) * w0;
            else val = fma(w3, t[g][3], fma(w2, t[g][2], fma(w1, t[g][1], w0 * t[g][0])));
            float out = (float)val;
            if (SLOW) {
                out = (dmask >> g) & 1 ? fillv : out;
                if (!nostore && l32 + 32 * g < ncol_valid) {
                    if (OUT == 3) qp.tok16[(o_off + qp.colbase + 32u * g) ^ 1u] = (unsigned short)quant16(out, qp);    /* token k = halfword k ^ 1 */
                    else if (OUT != 2) orow[32 * g] = out;
                    if (STATS) { vmin = fminf(vmin, out); vmax = fmaxf(vmax, out); }
                }
            } else {
                if (OUT != 2) __builtin_nontemporal_store(out, &orow[32 * g]);      /* streaming output (`nt`): -0.7 us per cfg2 field */
                if (STATS) { vmin = fminf(vmin, out); vmax = fmaxf(vmax, out); }
            }
        }
    }
}

/* y-pass of the token instantiation (OUT = 3), full strips without DEHORS columns: lane l32 owns the column PAIRS
 * (2 l32, 2 l32 + 1) + 64 g, g < 4, so the two 16-bit tokens of a stream word sit in one lane: per row pair 16
 * ds_read_b128 (the LDS cycles of the 32 ds_read_b64 of the float form) and 4 dword stores of 128 contiguous bytes per
 * row.  Same fma chains, same float rounding, then compact_float's quantisation (compact.tmplc:285-300). */
typedef double d2_t __attribute__((ext_vector_type(2)));
template <int DEG, int XR, bool QNT>
__device__ __forceinline__ void sepx_ypass_q(const float *myrec, const double *tcol2, unsigned *zw, const QuantP &qp)
{
#pragma unroll
    for (int h = 0; h < XR / 8; h++) {
        const float4 *r4 = (const float4 *)(myrec + h * 8 * 16);
        const float4 ra = r4[0], rb = r4[1], rc = r4[2];
        const unsigned o_off = (unsigned)__float_as_int(myrec[h * 8 * 16 + 12]);
        const double w0 = __hiloint2double(__float_as_int(ra.y), __float_as_int(ra.x));
        const double w1 = __hiloint2double(__float_as_int(ra.w), __float_as_int(ra.z));
        const double w2 = __hiloint2double(__float_as_int(rb.y), __float_as_int(rb.x));
        const double w3 = __hiloint2double(__float_as_int(rb.w), __float_as_int(rb.z));
        const char *tb = (const char *)tcol2;
        const unsigned a0 = lds_addr_of(tb + __float_as_int(rc.x)), a1 = lds_addr_of(tb + __float_as_int(rc.y));
        const unsigned a2 = lds_addr_of(tb + __float_as_int(rc.z)), a3 = lds_addr_of(tb + __float_as_int(rc.w));
        d2_t t[4][4];
#define RD4(A, J) asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\tds_read_b128 %2, %4 offset:1024\n\tds_read_b128 %3, %4 offset:1536" \
                               : "=&v"(t[0][J]), "=&v"(t[1][J]), "=&v"(t[2][J]), "=&v"(t[3][J]) : "v"(A) : "memory")
        RD4(a0, 0);
        if (DEG >= 1) RD4(a1, 1);
        if (DEG == 3) { RD4(a2, 2); RD4(a3, 3); }
#undef RD4
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (DEG == 3) asm volatile("" : "+v"(t[g][0]), "+v"(t[g][1]), "+v"(t[g][2]), "+v"(t[g][3]));
            else if (DEG == 1) asm volatile("" : "+v"(t[g][0]), "+v"(t[g][1]));
            else asm volatile("" : "+v"(t[g][0]));
        }
        unsigned *orow = zw + (o_off >> 1);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            unsigned tk[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                double val;
                if (DEG == 0) val = t[g][0][e];
                else if (DEG == 1) val = t[g][0][e] + (t[g][1][e] - t[g][0][e]) * w0;
                else val = fma(w3, t[g][3][e], fma(w2, t[g][2][e], fma(w1, t[g][1][e], w0 * t[g][0][e])));
                tk[e] = quant16((float)val, qp);
            }
            if (QNT) __builtin_nontemporal_store(tk[0] << 16 | tk[1], &orow[32 * g]); else orow[32 * g] = tk[0] << 16 | tk[1];
        }
    }
}

/* {min key, max key, 0} of the values a thread block stored -> triple `slot` of the field's partials (the layout
 * k_cf_header reduces: compact_float's min/max pass fused into the interpolation, STATS instantiation only) */
__device__ __forceinline__ void block_minmax_partial(float vmin, float vmax, unsigned *triple)
{
    for (int off = 32; off > 0; off >>= 1) {
        vmin = fminf(vmin, __shfl_down(vmin, off, 64));
        vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
    }
    __shared__ float shm[2][SEP_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) { shm[0][threadIdx.x >> 6] = vmin; shm[1][threadIdx.x >> 6] = vmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = fminf(fminf(shm[0][0], shm[0][1]), fminf(shm[0][2], shm[0][3]));
        float b = fmaxf(fmaxf(shm[1][0], shm[1][1]), fmaxf(shm[1][2], shm[1][3]));
        triple[0] = f2key(a); triple[1] = f2key(b); triple[2] = 0u;
    }
}

template <int DEG, int XR, int OUT>
__global__ __launch_bounds__(SEP_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 6)))
void k_sepx(ezhip_sep_plan p, float *__restrict__ zout, const float *__restrict__ zin)
{
    extern __shared__ __attribute__((aligned(16))) double smem_x[];
    constexpr bool STATS = OUT == 1 || OUT == 2;
    /* XCD-aware work mapping.  Thread blocks are dealt round-robin to the 8 XCDs in linear launch order, and every
     * XCD has its own L2.  Here XCD k takes the k-th CONTIGUOUS eighth of the (field, segment, strip) space, so the
     * strips that share source columns (35 of 192 staged floats) and the segments that share halo rows run on the
     * same L2: measured fabric reads 56 -> (see profiles) MB per cfg2 field. */
    /* 1-D launch: [pole_blocks producers][work items].  pole_blocks is a multiple of 8, so a work item's XCD is
     * (its index & 7) as well. */
    const unsigned L = blockIdx.x, P = p.pole_blocks;
    if (L < P) {
        /* producer of one pole value (ez_calcpoleval: a SEQUENTIAL REAL sum, ~20 us on one lane): dispatched first,
         * published with a release store of the launch epoch; the special-row blocks of the field (last in its
         * work order) acquire it.  Replaces a separate k_polevals launch (3 % of a cfg2 batch). */
        const unsigned f = L >> 1, which = L & 1;
        if ((int)f >= (p.batch_fields > 1 ? p.batch_fields : 1)) return;
        const float *zf = zin + f * p.batch_in_stride;
        const float *row = which == 0 ? zf + (size_t)(p.nj_src - 1) * p.ni_src : zf;
        const int chunk = (int)((p.x_lds_bytes / sizeof(float) - 4) & ~(size_t)3);
        const float v = block_poleval(row, p.ni_src, p.pole_weighted, p.ax, (float *)smem_x, chunk);
        if (threadIdx.x == 0) {
            __hip_atomic_store(&p.pole_gran[L], (unsigned long long)p.pole_epoch << 32 | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    int bx, by, bz;
    {
        const unsigned nx = p.x_nbx, nxy = nx * (unsigned)(p.by_cnt > 0 ? p.by_cnt : p.x_nseg + p.n_special);
        const unsigned total = nxy * (unsigned)(p.batch_fields > 1 ? p.batch_fields : 1);
        const unsigned M = L - P, full = total & ~7u;
        const unsigned w = M < full ? (M & 7u) * (full >> 3) + (M >> 3) : M;
        bz = w / nxy; const unsigned r = w - bz * nxy; by = r / nx; bx = r - by * nx;
        if (p.by_cnt > 0) by += p.by_lo;
        if (p.special_last == 1 && p.by_cnt == 0 && p.batch_fields <= 1 && p.n_special > 0) {
            /* a lone field: every XCD gets a contiguous eighth of the main (strip, segment) items AND an eighth of the special-row blocks
             * (inside the plain eighths the specials, last in the work order, all land on the last two XCDs, which then hold fewer main
             * blocks than the other six).  Worth 1 % (33.0 -> 32.7 us): every XCD still runs two rounds of main blocks */
            const unsigned nmain = nx * (unsigned)p.x_nseg, nspec = nx * (unsigned)p.n_special;
            const unsigned c = nmain >> 3, d = nspec >> 3, bal = 8u * (c + d);
            unsigned main_w = 0xFFFFFFFFu, spec_s = 0;
            if (M < bal) { const unsigned xcd = M & 7u, idx = M >> 3; if (idx < c) main_w = xcd * c + idx; else spec_s = xcd * d + (idx - c); }
            else { const unsigned t = M - bal, rm = nmain & 7u; if (t < rm) main_w = 8u * c + t; else spec_s = 8u * d + (t - rm); }
            if (main_w != 0xFFFFFFFFu) { by = main_w / nx; bx = main_w - by * nx; }
            else { const unsigned row = spec_s / nx; bx = spec_s - row * nx; by = p.x_nseg + row; }
            bz = 0;
        }
    }
    const int c = bx * SEP_BLOCK + threadIdx.x;
    const int cc = min(c, p.ni_dst - 1);
    const float fillv = p.fill ? *p.fill : 0.0f;
    /* bz = field of a batch launch (c_ezsint_batch_dev): no ramp-up / drain gap between fields */
    zin += bz * p.batch_in_stride; zout += bz * p.batch_out_stride;        /* OUT = 3: zout is the token array, its stride in words */
    QuantP qp; qp.minF = 0.0; qp.mul = 0.0; qp.tok16 = (unsigned short *)zout; qp.colbase = 0;
    if (OUT == 3) {
        const double *pp = (const double *)((const char *)p.quant_params + (size_t)bz * p.quant_stride);     /* packhip_cf_params: minF, mulFactor first */
        qp.minF = pp[0]; qp.mul = pp[1];
    }
    /* special rows sit in the MIDDLE of the field's work order: their gathers are latency-bound (tens of us per block),
     * so they must overlap main blocks (last in the order they became an exposed tail: +6 us per field), and by then
     * the pole producers have long finished */
    const int sp0 = p.special_last ? (p.special_last > 1 ? min(p.special_last - 2, p.x_nseg) : p.x_nseg) : p.x_nseg >> 1;      /* a lone field: last (nothing to overlap, and they must not hold slots while the pole producers run) */
    if (by >= sp0 && by < sp0 + p.n_special) {
        if (P) {
            p.pole_inline = 0;
            if (p.need_poles) {
                /* RELAXED agent-scope loads (sc1: served by the coherence point, never by a stale L1/L2 line), not
                 * acquire: on the 8-XCD part an agent-scope acquire is `buffer_inv sc1`, which drops the XCD's
                 * non-local L2 lines -- 232 special-row blocks per field doing that cost +6 us per field.  The flag
                 * load, the branch on it and the (equally cache-bypassing) value loads in sep_special issue in order. */
                p.pole_inline = 1;
                for (int k = 0; k < 2; k++) {
                    int spins = 0;
                    unsigned long long g;
                    /* value and epoch arrive in ONE 64-bit word: no ordering between two loads to rely on */
                    while ((unsigned)((g = __hip_atomic_load(&p.pole_gran[2 * bz + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != p.pole_epoch) {
                        __builtin_amdgcn_s_sleep(32);
                        if (++spins > (1 << 22)) {       /* seconds: never in a healthy launch.  Do not hang the device, and do not go on with
                                                            whatever the slot holds: the polar rows become NaN and the host learns of it */
                            p.pole_timeout = 1;
                            if (p.err_word && threadIdx.x == 0) __hip_atomic_store(p.err_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                    }
                    p.pole_now[k] = __uint_as_float((unsigned)g);
                }
            }
        } else if (p.polevals) p.polevals += 2 * bz;
        const float sv = sep_special<DEG, OUT>(p, zout, zin, by - sp0, c, cc, c < p.ni_dst, fillv, qp);
        if (STATS) block_minmax_partial(sv, sv, p.stat_partials + (size_t)bz * p.stat_stride + 3 * (size_t)(by * p.x_nbx + bx));
        return;
    }
    const int seg = by < sp0 ? by : by - p.n_special;
    const int i0 = seg * p.x_rb, i1 = min(i0 + p.x_rb, p.x_nvb);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nis = p.ni_src, nid = p.ni_dst, trows = p.x_tr, wstr = p.wstride;
    const int base = p.blk_base[bx], W = p.blk_w[bx];
    double *T = smem_x;                                                         /* T[slot][256 columns] */
    constexpr int SEPX_REC_DW = XR * 16;                                        /* dwords of one step's row records */
    float *rec = (float *)(smem_x + (size_t)trows * SEP_BLOCK);                /* 2 x XR row records of 64 B */
    float *patch = rec + 2 * SEPX_REC_DW;
    const int dbg = EZH_DBG(p.debug_flags);          /* development knock-outs (EZHIP_DEBUG): 1 no stores, 4 no DMA, 8 no x-pass, 16 no y-pass */
    unsigned coloff[SEP_QCH];
#pragma unroll
    for (int q = 0; q < SEP_QCH; q++) {          /* source column of patch column lane + 64 q (seam unrolled; tail lanes clamp) */
        int col = base + min(lane + 64 * q, W - 1);
        if (col >= nis) col -= nis;
        coloff[q] = (unsigned)col * 4u;
    }
    /* 16-byte DMA: lane l fetches patch columns 4 l .. 4 l + 3; usable when the window does not wrap at the seam, the
     * row holds it in at most 64 lanes and the last lane stays inside the source row */
    const int x4lanes = (W + 3) >> 2;
    const bool x4ok = !(dbg & 128) && base + 4 * x4lanes <= nis && x4lanes <= 64 && 4 * x4lanes <= wstr;
    const unsigned x4off = (unsigned)(base + 4 * min(lane, x4lanes - 1)) * 4u;
    const float *pcol = patch + p.coff[cc];
    const double cw[4] = {p.cw[cc], p.cw[nid + cc], p.cw[2 * nid + cc], p.cw[3 * nid + cc]};
    /* y-pass lane geometry: row r of the pair, column l32 + 32 g of the strip */
    const int l32 = lane & 31, rsub = lane >> 5;
    const int ncol_valid = min(SEP_BLOCK, nid - bx * SEP_BLOCK);   /* < 256 only in the last column block */
    const bool full = ncol_valid == SEP_BLOCK;
    /* DEHORS columns (extrapolation targets): bit g of dmask = column l32 + 32 g takes the fill value */
    unsigned dmask = 0;
    {
        __shared__ unsigned char cf[SEP_BLOCK];
        cf[threadIdx.x] = p.cflag[cc];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 8; g++) dmask |= (cf[l32 + 32 * g] != 0 ? 1u : 0u) << g;
    }
    const bool any_dehors = __syncthreads_or(dmask != 0) != 0;
    const bool slow = any_dehors || !full || (dbg & 1);      /* block-uniform: predicated y-pass */
    float *zcol = zout + (size_t)bx * SEP_BLOCK + l32;                 /* + row offset (record) + 32 g */
    qp.colbase = (unsigned)(bx * SEP_BLOCK + l32);

    auto load_step = [](const ezhip_xstep *tab, int i) {       /* four scalar loads (constant address space) */
        const auto *q = CONSTP(int, tab) + 4 * i;
        ezhip_xstep s; s.s0 = q[0]; s.n = q[1]; s.slot0 = q[2]; s.by = q[3];
        return s;
    };
    auto dma_issue = [&](const ezhip_xstep &st, int i) {
        if (dbg & 4) return;
        for (int row = wv; row < st.n; row += SEP_BLOCK / 64) {
            const float *zr = zin + (size_t)(st.s0 + row) * nis;
            float *prow = patch + row * wstr;
            if (x4ok) {          /* one 16-byte DMA per lane covers the whole patch row (no seam inside the window) */
                if (lane < x4lanes) lds_dma_dwordx4(zr, x4off, lds_addr_of(prow));
            } else {
#pragma unroll
                for (int q = 0; q < SEP_QCH; q++)
                    if (64 * q < W && lane + 64 * q < wstr) lds_dma_dword(zr, coloff[q], lds_addr_of(prow + 64 * q));
            }
        }
        /* the XR row records of step i: 16 XR dwords, one 64-dword chunk per wave */
        if (wv * 64 < SEPX_REC_DW)
            lds_dma_dword((const float *)(p.x_rows + (size_t)i * XR), (unsigned)(threadIdx.x * 4), lds_addr_of(rec + (i & 1) * SEPX_REC_DW + wv * 64));
    };
    float vmin = INFINITY, vmax = -INFINITY;                 /* STATS: extrema of the values this lane stores */
    ezhip_xstep st = load_step(p.x_first, i0);
    dma_issue(st, i0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = i0; i < i1; i++) {
        __syncthreads();                                   /* DMA(i) landed in every wave; everyone left y-pass(i-1) */
        /* ---- x-pass of the new source rows -> ring: chunks of four rows (16 LDS reads in flight per wait), then
         * chunks of two; an odd count repeats the last row */
        if (!(dbg & 8)) {
            int slot = st.slot0, s = 0;
            auto next_slot = [&](int sl) { sl++; return sl >= trows ? sl - trows : sl; };
            for (; s + 4 <= st.n; s += 4) {
                const float *r0 = pcol + s * wstr;
                const double t0 = xrow<DEG>(r0, cw), t1 = xrow<DEG>(r0 + wstr, cw), t2 = xrow<DEG>(r0 + 2 * wstr, cw), t3 = xrow<DEG>(r0 + 3 * wstr, cw);
                const int s1 = next_slot(slot), s2 = next_slot(s1), s3 = next_slot(s2);
                T[slot * SEP_BLOCK + threadIdx.x] = t0; T[s1 * SEP_BLOCK + threadIdx.x] = t1;
                T[s2 * SEP_BLOCK + threadIdx.x] = t2; T[s3 * SEP_BLOCK + threadIdx.x] = t3;
                slot = next_slot(s3);
            }
            for (; s < st.n; s += 2) {
                const float *r0 = pcol + s * wstr, *r1 = r0 + (s + 1 < st.n ? wstr : 0);
                const double t0 = xrow<DEG>(r0, cw), t1 = xrow<DEG>(r1, cw);
                const int s1 = next_slot(slot);
                T[slot * SEP_BLOCK + threadIdx.x] = t0;
                if (s + 1 < st.n) T[s1 * SEP_BLOCK + threadIdx.x] = t1;
                slot = next_slot(s1);
            }
        }
        __syncthreads();                                   /* the patch is free again; the ring holds window i for all columns */
        ezhip_xstep nst = st;
        if (i + 1 < i1) { nst = load_step(p.x_cont, i + 1); dma_issue(nst, i + 1); }
        /* ---- y-pass: wave wv owns target rows {2 wv, 2 wv + 1} and {8 + 2 wv, 9 + 2 wv} of the row-block */
        if (!(dbg & 16)) {
            const float *myrec = rec + (i & 1) * SEPX_REC_DW + (2 * wv + rsub) * 16;
            if (slow) sepx_ypass<DEG, XR, true, OUT>(myrec, T + l32, zcol, fillv, dmask, l32, ncol_valid, (dbg & 1) != 0, vmin, vmax, qp);
            else if (OUT == 3) { if (dbg & 256) sepx_ypass_q<DEG, XR, true>(myrec, T + 2 * l32, (unsigned *)zout + (size_t)bx * (SEP_BLOCK / 2) + l32, qp);
                                 else sepx_ypass_q<DEG, XR, false>(myrec, T + 2 * l32, (unsigned *)zout + (size_t)bx * (SEP_BLOCK / 2) + l32, qp); }
            else sepx_ypass<DEG, XR, false, OUT>(myrec, T + l32, zcol, fillv, dmask, l32, ncol_valid, false, vmin, vmax, qp);
        }
        st = nst;
        if ((dbg & 16) || slow || OUT == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* OUT = 2 stores nothing: only DMA(i+1) is outstanding */
        else if (i + 1 < i1) {                              /* DMA(i+1) landed; the stores of this step stay in flight */
            if (OUT == 3) { if (XR == 16) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else if (XR == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    if (STATS) block_minmax_partial(vmin, vmax, p.stat_partials + (size_t)bz * p.stat_stride + 3 * (size_t)(by * p.x_nbx + bx));
}

extern "C" size_t ezhip_sepx_lds_bytes(int x_tr, int rows_per_step, int x_prows, int wstride)
{
    size_t b = sizeof(double) * (size_t)x_tr * SEP_BLOCK + sizeof(float) * (2 * 16 * (size_t)rows_per_step + (size_t)x_prows * wstride);
    b += (size_t)EZH_DEVINT("EZHIP_SEPX_PAD");      /* development: occupancy experiments */
    return b;
}

/* rows per step: 16 (8 was measured too: smaller ring and patch, 4 blocks per CU, twice the barriers -- no gain) */
#define SEPX_DISPATCH(DEGV, OUTV, EXPR) do { \
        constexpr int X = 16; \
        const int o_ = (OUTV); \
        if ((DEGV) == 0) { constexpr int D = 0; if (o_ == 0) { constexpr int S = 0; EXPR; } else if (o_ == 1) { constexpr int S = 1; EXPR; } else if (o_ == 2) { constexpr int S = 2; EXPR; } else { constexpr int S = 3; EXPR; } } \
        else if ((DEGV) == 1) { constexpr int D = 1; if (o_ == 0) { constexpr int S = 0; EXPR; } else if (o_ == 1) { constexpr int S = 1; EXPR; } else if (o_ == 2) { constexpr int S = 2; EXPR; } else { constexpr int S = 3; EXPR; } } \
        else { constexpr int D = 3; if (o_ == 0) { constexpr int S = 0; EXPR; } else if (o_ == 1) { constexpr int S = 1; EXPR; } else if (o_ == 2) { constexpr int S = 2; EXPR; } else { constexpr int S = 3; EXPR; } } } while (0)
/* OUT of a plan: 1 when min/max partials are requested next to the float field */
static int plan_out(const ezhip_sep_plan *p) { return p->out_mode ? p->out_mode : (p->stat_partials ? 1 : 0); }

extern "C" int ezhip_sepx_capacity(int degree, int rows_per_step, size_t lds_bytes)
{
    int dev = 0, ncu = 0, nb = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    hipError_t e = hipSuccess;
    (void)rows_per_step;
    SEPX_DISPATCH(degree, 0, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sepx<D, X, S>, SEP_BLOCK, lds_bytes));
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nb * ncu;
}

static int launch_sepx(const ezhip_sep_plan *plan, float *d_zout, const float *d_zin)
{
    ezhip_sep_plan pl = *plan;
    plan = &pl;
    pl.err_word = pl.pole_blocks ? dev_err_word() : nullptr;
    pl.pole_timeout = getenv("EZHIP_TEST_POLE_TIMEOUT") ? 1 : 0;        /* tests: behave as if the wait had timed out */
    if (pl.pole_timeout && pl.err_word) __atomic_store_n(pl.err_word, 1, __ATOMIC_RELEASE);
    const size_t nf = plan->batch_fields > 1 ? plan->batch_fields : 1;
    pl.x_nbx = (plan->ni_dst + SEP_BLOCK - 1) / SEP_BLOCK;
    const size_t nblocks = (size_t)pl.pole_blocks + (size_t)pl.x_nbx * (size_t)(plan->by_cnt > 0 ? plan->by_cnt : plan->x_nseg + plan->n_special) * nf;
    if (nblocks >= ((size_t)1 << 31)) { snprintf(g_err, sizeof(g_err), "k_sepx: batch too large for one launch"); return -1; }
    dim3 grid((unsigned)nblocks), block(SEP_BLOCK);
    size_t lds = ezhip_sepx_lds_bytes(plan->x_tr, plan->x_rows_per_step, plan->x_prows, plan->wstride);
    if (pl.pole_blocks && lds < 4 * 1028) lds = 4 * 1028;          /* the pole producers stage >= 1024 floats per pass */
    pl.x_lds_bytes = lds;
    if (lds > 64 * 1024) {        /* tall windows on wide strips: raise the per-kernel dynamic LDS limit */
        hipError_t e = hipSuccess;
        SEPX_DISPATCH(plan->degree, plan_out(plan),
                      e = hipFuncSetAttribute((const void *)k_sepx<D, X, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      /* not the CU's 160 KB: the kernel's static LDS counts against it */
        if (e != hipSuccess) return set_err(e, "k_sepx LDS size");
    }
    if (plan_out(plan) == 3 && (plan->ni_dst & 1)) { snprintf(g_err, sizeof(g_err), "k_sepx: token output needs an even ni_dst"); return -1; }
    SEPX_DISPATCH(plan->degree, plan_out(plan), hipLaunchKernelGGL((k_sepx<D, X, S>), grid, block, lds, g_stream, *plan, d_zout, d_zin));
    return LAUNCH_CHECK("k_sepx");
}

extern "C" int ezhip_interp_sep(const ezhip_sep_plan *plan, float *d_zout, const float *d_zin)
{
    if (plan->x_nseg > 0) return launch_sepx(plan, d_zout, d_zin);
    /* fallback tile kernel: plans k_sepx cannot take (gather column blocks, source windows taller than the patch) */
    dim3 grid((plan->ni_dst + SEP_BLOCK - 1) / SEP_BLOCK, plan->nblk_y + plan->n_special);
    dim3 block(SEP_BLOCK);
    size_t lds = sizeof(float) * (size_t)plan->patch_elems;
    switch (plan->degree) {
    case 0: hipLaunchKernelGGL(k_sep<0>, grid, block, lds, g_stream, *plan, d_zout, d_zin); break;
    case 1: hipLaunchKernelGGL(k_sep<1>, grid, block, lds, g_stream, *plan, d_zout, d_zin); break;
    case 3: hipLaunchKernelGGL(k_sep<3>, grid, block, lds, g_stream, *plan, d_zout, d_zin); break;
    default: snprintf(g_err, sizeof(g_err), "ezhip_interp_sep: bad degree %d", plan->degree); return -1;
    }
    return LAUNCH_CHECK("k_sep");
}


/* ===================================================================================== */
/* k_sepx_enc : interpolate, quantise and armn_compress in ONE launch (cfg5 passes B + E)  */
/* ===================================================================================== */
/* Round 2's pipeline wrote compact_float's 16-bit tokens of every interpolated value to HBM (k_sepx<.., tokens>, 52 MB per cfg5 field) for
 * the encoder (k_armn_enc1, pack_kernels.hip) to read back: 104 of the pipeline's 234 MB per field, and a staging wait that was 44 % of an
 * encoder block's life.  Here the tokens never leave the CU:
 *   - a thread block owns 256 target columns x 16 target rows: a context column and a context row (the Lorenzo predictor of a tile looks one
 *     token up and one to the left, c_zfstlib.c:691-696) + 85 x 5 tiles of 3 x 3 = five CHUNKS of the stream (a chunk = the 85 tiles of one tile
 *     row of a strip; the stream runs tile row by tile row, c_zfstlib.c:722-768).  Strips advance by 255 columns, row groups by 15 rows: one
 *     column in 256 and one row in 16 are interpolated twice;
 *   - staging, x-pass and y-pass are k_sepx's (same tables, same fma chains, same REAL rounding, compact.tmplc:285-300 quantisation): the
 *     tokens land in an LDS patch of 16 x 256 halfwords; the rows that are not main rows (polar strips, pole rows) are evaluated by
 *     sep_special in place; no ring reuse between thread blocks (a row group brings all its ~13 source rows: 1.4 x the x-pass work of k_sepx);
 *   - tiles, widths, block scan, stream image in LDS, hand-off of the shared boundary words: k_armn_enc1's, per chunk.  A chunk's position comes
 *     from the same decoupled look-back over {aggregate | inclusive prefix} granules, one wave per chunk; the chunks of a row group depend on
 *     aggregates of the OTHER strips of the same row group (the stream interleaves them), which are published before anybody waits, by thread
 *     blocks launched next to this one: launch order = (row group, field, strip), so the lowest unfinished (row group, field) set is always
 *     resident or next in line on every XCD; all waits are bounded (abort flag -> the host redoes the batch with the two-kernel path);
 *   - row 0 and column 0 (the stream's prefix, c_zfstlib.c:712-721) go to a side array of ni + nj - 1 tokens per field; a small kernel
 *     (k_sepenc_prefix, pack_kernels.hip) assembles the prefix words and the word the prefix shares with chunk 0 afterwards.
 * Fields that turn out not to be compressible (zlng -1), to need the 5-bit width field (-2) or whose launch gave up are redone by the host
 * through the two-kernel path, which has the tokens.  Bit-identical records (tests/test_gpu_packers.py). */
#define SE_TPR 85                 /* tiles per chunk: 255 new columns of a strip */
#define SE_TROWS 5                /* tile rows (chunks) per thread block: 15 new rows */
#define SE_TOKW 132               /* words per row of the LDS token patch: 256 tokens + 4 words (the two rows a wave stores at once land in different banks) */
#define SE_TPT 2                  /* tiles per thread: 425 tiles per thread block */
#define SE_IMG_WORDS ((SE_TPR * SE_TROWS * 167 + 31) / 32 + 3 * SE_TROWS + 8)   /* worst case: 5 + 9 x 18 bits per tile */

extern "C" size_t ezhip_sepenc_lds_bytes(const ezhip_sep_plan *p)
{
    size_t front = sizeof(double) * (size_t)p->e_tr * SEP_BLOCK + sizeof(float) * (16 * 16 + (size_t)p->e_prows * p->e_wstride);
    if (front < 4 * (size_t)SE_IMG_WORDS) front = 4 * (size_t)SE_IMG_WORDS;      /* the stream image takes the place of ring, records and patch */
    front = (front + 15) & ~(size_t)15;
    return front + 4 * 16 * (size_t)SE_TOKW;
}

/* y-pass of one wave into the LDS token patch: lane l32 owns the column pairs (2 l32, 2 l32 + 1) + 64 g of target rows {2 wv + rsub, 8 + 2 wv + rsub}
 * (sepx_ypass_q's lane geometry: 16 ds_read_b128 per row pair); tokens in natural order (column c = halfword c of the row) */
template <int DEG>
__device__ __forceinline__ void sepx_ypass_lds(const float *myrec, const double *tcol2, unsigned *trow, const QuantP &qp)
{
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const float4 *r4 = (const float4 *)(myrec + h * 8 * 16);
        const float4 ra = r4[0], rb = r4[1], rc = r4[2];
        const double w0 = __hiloint2double(__float_as_int(ra.y), __float_as_int(ra.x));
        const double w1 = __hiloint2double(__float_as_int(ra.w), __float_as_int(ra.z));
        const double w2 = __hiloint2double(__float_as_int(rb.y), __float_as_int(rb.x));
        const double w3 = __hiloint2double(__float_as_int(rb.w), __float_as_int(rb.z));
        const char *tb = (const char *)tcol2;
        const unsigned a0 = lds_addr_of(tb + __float_as_int(rc.x)), a1 = lds_addr_of(tb + __float_as_int(rc.y));
        const unsigned a2 = lds_addr_of(tb + __float_as_int(rc.z)), a3 = lds_addr_of(tb + __float_as_int(rc.w));
        d2_t t[4][4];
#define RD4(A, J) asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\tds_read_b128 %2, %4 offset:1024\n\tds_read_b128 %3, %4 offset:1536" \
                               : "=&v"(t[0][J]), "=&v"(t[1][J]), "=&v"(t[2][J]), "=&v"(t[3][J]) : "v"(A) : "memory")
        RD4(a0, 0);
        if (DEG >= 1) RD4(a1, 1);
        if (DEG == 3) { RD4(a2, 2); RD4(a3, 3); }
#undef RD4
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (DEG == 3) asm volatile("" : "+v"(t[g][0]), "+v"(t[g][1]), "+v"(t[g][2]), "+v"(t[g][3]));
            else if (DEG == 1) asm volatile("" : "+v"(t[g][0]), "+v"(t[g][1]));
            else asm volatile("" : "+v"(t[g][0]));
        }
        unsigned *orow = trow + h * 8 * SE_TOKW;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            unsigned tk[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                double val;
                if (DEG == 0) val = t[g][0][e];
                else if (DEG == 1) val = t[g][0][e] + (t[g][1][e] - t[g][0][e]) * w0;
                else val = fma(w3, t[g][3][e], fma(w2, t[g][2][e], fma(w1, t[g][1][e], w0 * t[g][0][e])));
                tk[e] = quant16((float)val, qp);
            }
            orow[32 * g] = tk[0] | tk[1] << 16;
        }
    }
}

template <int DEG>
__global__ __launch_bounds__(SEP_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 6)))
void k_sepx_enc(ezhip_sep_plan p, ezhip_sepenc_args a)
{
    extern __shared__ __attribute__((aligned(16))) double smem_x[];
    __shared__ unsigned s_wsum[SE_TPT][SEP_BLOCK / 64], s_rb[SE_TROWS + 1], s_ib[SE_TROWS + 1], s_gt, s_abort, s_gtall[SE_TROWS];
    __shared__ unsigned long long s_start[SE_TROWS];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    /* launch order: (row group, field, strip) -- the strips of a (row group, field) set are neighbours in the launch */
    const unsigned ns = (unsigned)p.e_nstrips, set = blockIdx.x / ns;
    const int s = (int)(blockIdx.x - set * ns), rg = (int)(set / (unsigned)a.nfields), f = (int)(set - (unsigned)rg * (unsigned)a.nfields);
    const int nis = p.ni_src, nid = p.ni_dst, njd = p.nj_dst, trows = p.e_tr, wstr = p.e_wstride;
    const int c = s * (3 * SE_TPR) + tid, cc = min(c, nid - 1);
    const float *zin = a.zin + (size_t)f * a.in_stride;
    QuantP qp; qp.tok16 = nullptr; qp.colbase = 0;
    { const double *pp = (const double *)((const char *)a.quant_params + (size_t)f * a.quant_stride); qp.minF = pp[0]; qp.mul = pp[1]; }
    double *T = smem_x;
    float *rec = (float *)(smem_x + (size_t)trows * SEP_BLOCK);
    float *patch = rec + 16 * 16;
    unsigned *tokw = (unsigned *)((char *)smem_x + p.x_lds_bytes);          /* set by the launcher: the bytes in front of the token patch */
    if (tid == 0) { s_gt = 0; s_abort = 0; }
    /* ---- interpolation: one step of k_sepx whose source rows are all new -------------------------------------------------- */
    const auto *sq = CONSTP(int, p.e_step) + 4 * rg;
    const int st_s0 = sq[0], st_n = sq[1], st_slot0 = sq[2];
    if (st_n > 0 && !(EZH_DBG(a.debug) & 64)) {
        const int base = p.e_blk_base[s], W = p.e_blk_w[s];
        unsigned coloff[SEP_QCH];
#pragma unroll
        for (int q = 0; q < SEP_QCH; q++) {
            int col = base + min(lane + 64 * q, W - 1);
            if (col >= nis) col -= nis;
            coloff[q] = (unsigned)col * 4u;
        }
        const int x4lanes = (W + 3) >> 2;
        const bool x4ok = base + 4 * x4lanes <= nis && x4lanes <= 64 && 4 * x4lanes <= wstr;
        const unsigned x4off = (unsigned)(base + 4 * min(lane, x4lanes - 1)) * 4u;
        for (int row = wv; row < st_n; row += SEP_BLOCK / 64) {
            const float *zr = zin + (size_t)(st_s0 + row) * nis;
            float *prow = patch + row * wstr;
            if (x4ok) { if (lane < x4lanes) lds_dma_dwordx4(zr, x4off, lds_addr_of(prow)); }
            else {
#pragma unroll
                for (int q = 0; q < SEP_QCH; q++)
                    if (64 * q < W && lane + 64 * q < wstr) lds_dma_dword(zr, coloff[q], lds_addr_of(prow + 64 * q));
            }
        }
        lds_dma_dword((const float *)(p.e_rows + (size_t)rg * 16), (unsigned)(threadIdx.x * 4), lds_addr_of(rec + wv * 64));      /* 16 records of 16 dwords */
        int off0 = p.cidx[cc] - base;
        if (off0 < 0) off0 += nis;
        const float *pcol = patch + off0;
        const double cw[4] = {p.cw[cc], p.cw[nid + cc], p.cw[2 * nid + cc], p.cw[3 * nid + cc]};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        {
            int slot = st_slot0, sr = 0;
            auto next_slot = [&](int sl) { sl++; return sl >= trows ? sl - trows : sl; };
            for (; sr + 4 <= st_n; sr += 4) {
                const float *r0 = pcol + sr * wstr;
                const double t0 = xrow<DEG>(r0, cw), t1 = xrow<DEG>(r0 + wstr, cw), t2 = xrow<DEG>(r0 + 2 * wstr, cw), t3 = xrow<DEG>(r0 + 3 * wstr, cw);
                const int s1 = next_slot(slot), s2 = next_slot(s1), s3 = next_slot(s2);
                T[slot * SEP_BLOCK + threadIdx.x] = t0; T[s1 * SEP_BLOCK + threadIdx.x] = t1;
                T[s2 * SEP_BLOCK + threadIdx.x] = t2; T[s3 * SEP_BLOCK + threadIdx.x] = t3;
                slot = next_slot(s3);
            }
            for (; sr < st_n; sr += 2) {
                const float *r0 = pcol + sr * wstr, *r1 = r0 + (sr + 1 < st_n ? wstr : 0);
                const double t0 = xrow<DEG>(r0, cw), t1 = xrow<DEG>(r1, cw);
                const int s1 = next_slot(slot);
                T[slot * SEP_BLOCK + threadIdx.x] = t0;
                if (sr + 1 < st_n) T[s1 * SEP_BLOCK + threadIdx.x] = t1;
                slot = next_slot(s1);
            }
        }
        __syncthreads();
        const int l32 = lane & 31, rsub = lane >> 5;
        sepx_ypass_lds<DEG>(rec + (2 * wv + rsub) * 16, T + 2 * l32, tokw + (2 * wv + rsub) * SE_TOKW + l32, qp);
    }
    __syncthreads();
    unsigned short *p16 = (unsigned short *)tokw;
    constexpr unsigned RP = 2 * SE_TOKW;                      /* halfwords per patch row */
    /* ---- the rows that are not main rows (polar strips, pole rows): one value per thread and row, in place ---------------- */
    {
        p.polevals = a.poles ? a.poles + 2 * f : nullptr; p.pole_inline = 0; p.pole_timeout = 0;
        const auto *es = CONSTP(int, p.e_special) + 16 * rg;
        bool any = false;
        for (int k = 0; k < 16; k++) {
            const int si = es[k];                              /* block-uniform */
            if (si < 0) continue;
            const float v = sep_special<DEG, 2>(p, nullptr, zin, si, c, cc, false, 0.0f, qp);
            p16[k * RP + tid] = (unsigned short)quant16(v, qp);
            any = true;
        }
        if (any) __syncthreads();
    }
    /* ---- the stream's prefix tokens: row 0 (first row group), column 0 (first strip) ------------------------------------- */
    {
        unsigned short *pt = a.ptok + (size_t)f * a.ptok_stride;
        if (rg == 0 && c < nid && (tid > 0 || s == 0)) pt[c] = p16[tid];
        if (s == 0 && tid >= 1 && tid < 16) { const int r = 3 * SE_TROWS * rg + tid; if (r < njd) pt[nid + r - 1] = p16[tid * RP]; }
    }
    if (EZH_DBG(a.debug) & 32) { if (p16[tid] == 0x1234 && p16[15 * RP + tid] == 0x4321) a.zlng[f] = 7; return; }      /* development: interpolation only */
    /* ---- per tile: differences (kept in registers), width, bit count (k_armn_enc1's tile phase on the LDS patch) ---------- */
    const int nbits = a.nbits, container = a.container;
    const int nt_x = min(SE_TPR, a.ntx - SE_TPR * s), nrow_t = min(SE_TROWS, a.nty - SE_TROWS * rg), ntl = nt_x * nrow_t;
    unsigned long long dpk[SE_TPT][3];
    unsigned bits[SE_TPT], meta[SE_TPT];                    /* meta: need | tm << 8 | tn << 12 | trow << 16 */
    bool gt = false;
#pragma unroll
    for (int q = 0; q < SE_TPT; q++) {
        const int tl = tid + SEP_BLOCK * q;
        bits[q] = 0; meta[q] = 0;
        dpk[q][0] = dpk[q][1] = dpk[q][2] = 0ull;
        if (tl < ntl) {
            const int trow = tl / nt_x, tcx = tl - trow * nt_x;
            const int tm = min(3, nid - (1 + 3 * (SE_TPR * s + tcx))), tn = min(3, njd - (1 + 3 * (SE_TROWS * rg + trow)));
            int h[4][3];
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const unsigned short *prow = p16 + (unsigned)(3 * trow + n) * RP + 3 * tcx;
                const int u0 = prow[0], u1 = prow[1], u2 = prow[2], u3 = prow[3];
                h[n][0] = u1 - u0; h[n][1] = u2 - u1; h[n][2] = u3 - u2;
            }
            int d[3][3];
#pragma unroll
            for (int n = 0; n < 3; n++)
#pragma unroll
                for (int m = 0; m < 3; m++) d[n][m] = h[n + 1][m] - h[n][m];        /* u11 - (u01 + u10 - u00), c_zfstlib.c:691-696 */
            if (tm < 3 || tn < 3) {
#pragma unroll
                for (int n = 0; n < 3; n++)
#pragma unroll
                    for (int m = 0; m < 3; m++) if (n >= tn || m >= tm) d[n][m] = 0;
            }
            int hi = max(max(d[0][0], d[0][1]), d[0][2]), lo = min(min(d[0][0], d[0][1]), d[0][2]);
#pragma unroll
            for (int n = 1; n < 3; n++) { hi = max(max(hi, d[n][0]), max(d[n][1], d[n][2])); lo = min(min(lo, d[n][0]), min(d[n][1], d[n][2])); }
            const int mx = max(hi, -lo);
#pragma unroll
            for (int n = 0; n < 3; n++) {
                const unsigned a0 = (unsigned)d[n][0] & 0x3FFFFu, a1 = (unsigned)d[n][1] & 0x3FFFFu, a2 = (unsigned)d[n][2] & 0x3FFFFu;
                dpk[q][n] = (unsigned long long)(a0 | a1 << 21) | (unsigned long long)(a1 >> 11 | a2 << 10) << 32;
            }
            if (mx > 65535) gt = true;
            unsigned need = (unsigned)bitlen((unsigned)mx);
            if (need == 16) need = 15;
            bits[q] = tile_bits(1, need, tm * tn, container, nbits);
            meta[q] = need | (unsigned)tm << 8 | (unsigned)tn << 12 | (unsigned)trow << 16;
        }
    }
    if (gt) s_gt = 1;
    /* ---- block scan over the tiles in (tile row, tile) order: ONE wave scan, the two layers in the halves of a word (<= 10 944 per wave) ---- */
    unsigned incl[SE_TPT];
    {
        static_assert(SE_TPT == 2, "the packed scan holds two layers");
        unsigned v = bits[0] | bits[1] << 16;
        for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(v, off, 64); if (lane >= off) v += o; }
        incl[0] = v & 0xFFFFu; incl[1] = v >> 16;
        if (lane == 63) { s_wsum[0][wv] = incl[0]; s_wsum[1][wv] = incl[1]; }
    }
    __syncthreads();                                        /* also: every thread is done with the token patch and s_gt is final */
    unsigned excl[SE_TPT], agg_all = 0;
#pragma unroll
    for (int q = 0; q < SE_TPT; q++) {
        unsigned before = agg_all;
        for (int w = 0; w < SEP_BLOCK / 64; w++) { if (w < wv) before += s_wsum[q][w]; agg_all += s_wsum[q][w]; }
        excl[q] = before + incl[q] - bits[q];
    }
#pragma unroll
    for (int q = 0; q < SE_TPT; q++) {                      /* the first tile of every chunk: the chunk's base in the block's scan */
        const int tl = tid + SEP_BLOCK * q;
        if (tl < ntl) { const int trow = (int)(meta[q] >> 16); if (tl == trow * nt_x) s_rb[trow] = excl[q]; }
    }
    if (tid == 0) s_rb[nrow_t] = agg_all;
    __syncthreads();
    const int ty0 = SE_TROWS * rg;
    unsigned long long *status = a.status + (size_t)f * a.nchunks, *tail = a.tail + (size_t)f * a.nchunks;
    if (tid < nrow_t) {
        const int cr = (ty0 + tid) * (int)ns + s;
        st_granule(&status[cr], (cr == 0 ? ST_PFX : ST_AGG) | (s_gt ? ST_GT : 0ull) | (unsigned long long)(s_rb[tid + 1] - s_rb[tid]));      /* chunk 0: its aggregate IS its inclusive prefix */
    }
    /* image word offsets of the chunks: every chunk starts on a word, two spare words behind it */
    unsigned rbase[SE_TROWS + 1], ib[SE_TROWS + 1];
    ib[0] = 0;
#pragma unroll
    for (int r = 0; r <= SE_TROWS; r++) rbase[r] = s_rb[min(r, nrow_t)];
#pragma unroll
    for (int r = 0; r < SE_TROWS; r++) ib[r + 1] = ib[r] + ((rbase[r + 1] - rbase[r] + 31) >> 5) + 2;
    unsigned *img = (unsigned *)smem_x;
    for (unsigned w = tid; w < ib[SE_TROWS] + 4; w += SEP_BLOCK) img[w] = 0;
    if (tid <= SE_TROWS) { unsigned v = ib[0]; for (int r = 1; r <= SE_TROWS; r++) if (tid == r) v = ib[r]; s_ib[tid] = v; }      /* for the loops over a runtime chunk index below */
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SE_TPT; q++) {
        if (tid + SEP_BLOCK * q >= ntl || (EZH_DBG(a.debug) & 1)) continue;
        const unsigned need = meta[q] & 0xFF; const int tm = (int)(meta[q] >> 8) & 0xF, tn = (int)(meta[q] >> 12) & 0xF, trow = (int)(meta[q] >> 16);
        unsigned rb_ = rbase[0], ib_ = ib[0];
#pragma unroll
        for (int r = 1; r < SE_TROWS; r++) if (trow == r) { rb_ = rbase[r]; ib_ = ib[r]; }
        const unsigned pos = excl[q] - rb_;
        unsigned wi = ib_ + (pos >> 5); int fill = (int)(pos & 31);
        unsigned long long acc = 0; bool first = true;
        const int width = need == 0 ? 0 : (need == 15 ? 17 : (int)need + 1);
        const unsigned mask = (1u << width) - 1;
        const int rowlen = tm * width;
        auto flush = [&]() {
            const unsigned word = (unsigned)(acc >> 32);
            if (first) { atomicOr(&img[wi], word); first = false; } else img[wi] = word;
            wi++; acc <<= 32; fill -= 32;
        };
#pragma unroll
        for (int n = 0; n < 3; n++) {
            if (n >= tn || (n > 0 && need == 0)) break;
            unsigned long long v = 0;
#pragma unroll
            for (int m = 0; m < 3; m++) if (m < tm) v = (v << width) | (unsigned long long)((unsigned)(dpk[q][n] >> (21 * m)) & mask);
            int L = rowlen;
            if (n == 0) { v |= (unsigned long long)need << rowlen; L += container; }
            const int room = 64 - fill;
            if (L <= room) { acc |= L == 64 ? v : v << (room - L); fill += L; }
            else { acc |= v >> (L - room); fill = 64; flush(); flush(); acc = v << (64 - (L - room)); fill = L - room; continue; }
            if (fill >= 32) flush();
            if (fill >= 32) flush();
        }
        if (fill > 0) atomicOr(&img[wi], (unsigned)(acc >> 32));
    }
    __syncthreads();                                        /* the images are complete */
    /* ---- per chunk, one wave: publish the tail, look back for the chunk's position, settle the word shared with the earlier chunks ---- */
    const unsigned long long body_start = 32ull + 3ull + (unsigned long long)(nid + njd - 1) * (unsigned long long)nbits;
    for (int r = wv; r < nrow_t; r += SEP_BLOCK / 64) {
        const int cr = (ty0 + r) * (int)ns + s;
        const unsigned agg = s_rb[r + 1] - s_rb[r];
        const unsigned *im = img + s_ib[r];
        auto wait_tail = [&](int idx, unsigned long long t) -> unsigned {
            int spins = 0;
            while ((t >> 62) == 0) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 19) || ((spins & 63) == 0 && __hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(&a.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1; break;
                }
                t = ld_granule(&tail[idx]);
            }
            return (unsigned)t;
        };
        unsigned long long tprev = (lane == 0 && cr > 0) ? ld_granule(&tail[cr - 1]) : 0ull;
        if (lane == 0) {
            unsigned tl32;
            if (agg >= 32) { const unsigned e = agg & 31, w = agg >> 5; tl32 = e ? (im[w - 1] << e) | (im[w] >> (32 - e)) : im[w - 1]; }
            else {                                               /* a chunk shorter than a word (the few tiles of a last strip): chained.  Chunk 0 has 85 tiles */
                const unsigned tp = cr > 0 ? wait_tail(cr - 1, tprev) : 0u;
                tl32 = agg ? (tp << agg) | (im[0] >> (32 - agg)) : tp;
                tprev = ST_PFX | tp;
            }
            st_granule(&tail[cr], ST_PFX | (unsigned long long)tl32);
        }
        unsigned long long excl_chunks = 0, gt_before = 0;
        bool gave_up = false;
        if (cr > 0 && !(EZH_DBG(a.debug) & 8)) {
            int basei = cr - 1, spins = 0;
            for (;;) {
                const int idx = basei - lane;
                const unsigned long long stv = idx >= 0 ? ld_granule(&status[idx]) : ST_PFX;
                const unsigned long long pm = __ballot((stv >> 62) == 2), okm = __ballot((stv >> 62) != 0);
                const int firstp = pm ? __builtin_ctzll(pm) : 63;
                const unsigned long long needm = firstp == 63 ? ~0ull : ((2ull << firstp) - 1);
                if ((okm & needm) != needm) {
                    /* not there yet: ONE lane polls the ONE nearest granule that is missing, with a pause, then the window is read again */
                    const int bad_idx = basei - __builtin_ctzll(~okm & needm);
                    if (lane == 0) {
                        while ((ld_granule(&status[bad_idx]) >> 62) == 0) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > (1 << 19) || ((spins & 63) == 0 && __hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { gave_up = true; break; }
                        }
                    }
                    gave_up = __shfl((int)gave_up, 0, 64) != 0;
                    if (gave_up) break;
                    continue;
                }
                unsigned long long v = 0; bool gg = false;
                if (!pm || lane <= firstp) { v = ST_VAL(stv); gg = (stv & ST_GT) != 0; }
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                excl_chunks += v;
                gt_before |= __ballot(gg);
                if (pm) break;
                basei -= 64;
            }
            if (lane == 0 && !gave_up) st_granule(&status[cr], ST_PFX | ((gt_before || s_gt) ? ST_GT : 0ull) | (excl_chunks + agg));
        }
        if (lane == 0) {
            const unsigned long long S = body_start + excl_chunks;
            s_start[r] = S;
            s_gtall[r] = (gt_before || s_gt) ? 1u : 0u;
            if (gave_up) { s_abort = 1; __hip_atomic_store(&a.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else {
                const unsigned sh = (unsigned)(S & 31);
                if (sh && (((sh + agg) >> 5) >= 1 || cr == a.nchunks - 1) && !(EZH_DBG(a.debug) & 8)) {         /* the shared word is completed here (else a later chunk stores it) */
                    if (cr == 0) a.head[f] = im[0];                                   /* the prefix's bits come from the side array: k_sepenc_prefix stores the word */
                    else {
                        const unsigned tp = wait_tail(cr - 1, tprev);
                        const unsigned v0 = (im[0] >> sh) | ((tp & ((1u << sh) - 1u)) << (32 - sh));
                        if ((S >> 5) < a.z_cap) a.z[(size_t)f * a.z_stride + (S >> 5)] = v0;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (s_abort) return;
    /* ---- copy out: every chunk's image shifted to its absolute bit position ------------------------------------------------ */
    unsigned *z = a.z + (size_t)f * a.z_stride;
    for (int r = 0; r < nrow_t; r++) {
        const int cr = (ty0 + r) * (int)ns + s;
        const unsigned agg = s_rb[r + 1] - s_rb[r];
        const unsigned *im = img + s_ib[r];
        const unsigned long long S = s_start[r];
        const unsigned sh = (unsigned)(S & 31);
        const unsigned long long gw0 = S >> 5;
        const unsigned nwout = (sh + agg + 31) >> 5;
        const bool last_chunk = cr == a.nchunks - 1;
        const bool tail_open = ((sh + agg) & 31) != 0 && !last_chunk;
        const unsigned nstore = nwout - (tail_open ? 1u : 0u);
        if (!(EZH_DBG(a.debug) & 2))
        for (unsigned k = (sh ? 1u : 0u) + tid; k < nstore; k += SEP_BLOCK) {
            const unsigned lo = im[k];
            const unsigned v = sh == 0 ? lo : ((k ? im[k - 1] : 0u) << (32 - sh)) | (lo >> sh);
            if (gw0 + k < a.z_cap) z[gw0 + k] = v;
        }
        if (tid == 0 && last_chunk) {
            if (gw0 + nwout < a.z_cap) z[gw0 + nwout] = 0u;
            const unsigned long long bits_total = S + agg - 32;       /* c_zfstlib.c:160-179 */
            const long long zl = 1 + 4 * (1 + (long long)((bits_total + 31) / 32));
            const long long lng_origin = 1 + 2 * (long long)nid * njd;
            a.zlng[f] = (s_gtall[r] && container == 4 && nbits >= 15) ? -2 : (zl >= lng_origin ? -1 : (int)zl);
        }
    }
}

extern "C" int ezhip_interp_sep_enc(const ezhip_sep_plan *plan, const ezhip_sepenc_args *args)
{
    if (!plan->e_ok) { snprintf(g_err, sizeof(g_err), "k_sepx_enc: the plan has no fused geometry"); return -1; }
    ezhip_sep_plan pl = *plan;
    const size_t lds = ezhip_sepenc_lds_bytes(plan);
    pl.x_lds_bytes = lds - 4 * 16 * (size_t)SE_TOKW;
    const size_t nblocks = (size_t)pl.e_nstrips * (size_t)pl.e_nrg * (size_t)args->nfields;
    if (nblocks >= ((size_t)1 << 31)) { snprintf(g_err, sizeof(g_err), "k_sepx_enc: batch too large for one launch"); return -1; }
    hipError_t e = hipSuccess;
    const void *fn = pl.degree == 0 ? (const void *)k_sepx_enc<0> : pl.degree == 1 ? (const void *)k_sepx_enc<1> : (const void *)k_sepx_enc<3>;
    if (lds > 64 * 1024) { e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return set_err(e, "k_sepx_enc LDS size"); }
    dim3 grid((unsigned)nblocks), block(SEP_BLOCK);
    switch (pl.degree) {
    case 0: hipLaunchKernelGGL(k_sepx_enc<0>, grid, block, lds, g_stream, pl, *args); break;
    case 1: hipLaunchKernelGGL(k_sepx_enc<1>, grid, block, lds, g_stream, pl, *args); break;
    default: hipLaunchKernelGGL(k_sepx_enc<3>, grid, block, lds, g_stream, pl, *args); break;
    }
    return LAUNCH_CHECK("k_sepx_enc");
}


/* ===================================================================================== */
/* exact extrema of the interpolated field WITHOUT interpolating it (cfg5 pipeline, pass A) */
/* ===================================================================================== */
/* compact_float needs the minimum and the maximum of the interpolated field before the first token can be formed
 * (compact.tmplc:173-204); round 2 got them from a whole k_sepx pass that stored nothing (23 us per cfg2 field, bound by the
 * REAL*8 arithmetic of 25.9 M points).  But a value of the separable kernels is sum_j wy_j (sum_k wx_k z_jk) over ONE window of
 * ntap x ntap source points with sum w = 1: it cannot leave [wmin - a R, wmax + a R] of its window (R = wmax - wmin,
 * a = (max sum |wx| * max sum |wy| - 1) / 2 = 0.28 for the cubic weights; + a rounding allowance, ezhip_sep_plan.bb_s).  So
 *   k_bb_bounds  streams the SOURCE field once (9.7 M points instead of 25.9 M, min / max only): per window the interval, per tile of
 *                252 x 32 windows its extremes, and the best GUARANTEED value L = max over windows that hold a target point of
 *                their lower end (some value of the field is >= L), likewise U for the minimum;
 *   k_bb_reduce  L, U of the field; how many tiles can still hold the extremum;
 *   k_bb_eval    a window whose upper end is below L cannot hold the maximum (monotone float rounding keeps the order): only the
 *                windows that reach L (or U) -- a few around the field's extremes -- have their target points evaluated, with
 *                exactly the arithmetic of k_sep / k_sepx (same tables, same fma chains): the extrema are the ones the
 *                interpolating pass finds, bit for bit (tests/test_gpu_packers.py);
 *   k_bb_special the polar rows (special rows of the plan) are evaluated whole.
 * Windows with a non-finite value are always evaluated.  A field where more than max_cand tiles qualify (no extremum stands out:
 * a field of noise) is flagged instead and the caller runs the interpolating pass for it. */
#define BB_TW 252          /* window columns of a strip: the 64 lanes of a wave hold 256 consecutive source columns */
#define BB_TH 64           /* window rows of a tile (one wave sweeps them; 32: 9 % of the rows are read twice as halo, FETCH_SIZE 56 MB per 38.7 MB field) */
#define BB_SUB 8           /* window rows per block of the second sweep */
#define BB_TILE_F 8            /* floats per tile record */
#define BB_LIST_CAP (1 << 18)   /* candidate windows per field; more: the field is handed back (flags) */
struct bb_args {
    ezhip_sep_plan p;
    const float *zin; size_t in_stride; int nfields;
    int ntx, nty;                    /* tiles per field */
    float sf;                        /* bb_s rounded up to REAL with room for the REAL arithmetic of the bounds themselves */
    float *tile;                     /* [nfields][nty * ntx][4]: max upper end, min lower end, max lower end (-> L), min upper end (-> U) over the tile's occupied windows */
    unsigned *tile_bad;              /* [nfields][nty * ntx]: the tile holds a non-finite value: every window of it is evaluated, none contributes to L / U */
    float *LU;                       /* [nfields][2] */
    unsigned *count, *list;          /* [nfields], [nfields][BB_LIST_CAP]: windows to evaluate, i0 | j0 << 16 */
    unsigned *ntl, *tlist;           /* [nfields], [nfields][nty * ntx]: the tiles that qualify (k_bb_reduce), swept by k_bb_select */
    unsigned *keys; size_t key_stride;   /* [f * key_stride + 0..2] = {min key, max key, 0} */
    int *flags; int max_cand, force_all, list_cap, exact_ok;
    const float *poles;
};

/* One wave sweeps the windows [tx BB_TW, +BB_TW) x [ty BB_TH, +BB_TH) of field zf: visit(valid, i0, j0, ub, lb) in wave-uniform control flow for every
 * window row that holds a target row; valid = the window holds a target column.  COARSE = false: four calls, the lane's four windows; COARSE = true: ONE
 * call for the lane's four windows together (the extremes of its 7 x NT values bound each of them: looser, a quarter of the arithmetic -- the first
 * pass only needs the tile's extremes and L / U, and a window that qualifies by its own bounds sits in a tile that qualifies by these).
 * The minima / maxima run on order-preserving INTEGER keys of the REAL bit patterns (k = u ^ ((u >> 31) & 0x7fffffff), signed compare): v_min_i32
 * needs no canonicalisation of its inputs (fminf costs three instructions on raw loads), and a NaN or an infinity becomes the largest or smallest
 * key instead of being skipped: *kmin / *kmax (the lane's extreme keys) tell the caller whether the tile holds a non-finite value.
 * Sliding extremes: horizontally the six pair extremes of the lane's seven values serve its four windows, vertically the pair (row r - 1, row r)
 * serves the windows that end at r and at r + 2. */
__device__ __forceinline__ int bb_key(float v) { const int u = __float_as_int(v); return u ^ ((u >> 31) & 0x7fffffff); }
__device__ __forceinline__ float bb_unkey(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
#define BB_KEY_POS_INF 0x7f800000            /* keys >= this: +inf / NaN; keys <= ~this: -inf / -NaN */
template <int NT, bool VEC, bool COARSE, class F>
__device__ __forceinline__ void bb_sweep(const bb_args &a, const float *zf, int tx, int j0a, int j0b_, int *kmin_out, int *kmax_out, F &&visit)
{
    const int lane = threadIdx.x & 63;
    const int ni = a.p.ni_src, nj = a.p.nj_src;
    const int nwr = nj - NT + 1;
    const int j0b = min(j0b_, nwr);                           /* window rows [j0a, j0b) */
    int kmn = 0x7fffffff, kmx = (int)0x80000000;
    *kmin_out = kmn; *kmax_out = kmx;
    if (j0a >= j0b) return;
    int col[4]; bool own[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int w = 4 * lane + q, cu = tx * BB_TW + w;
        col[q] = cu % ni;
        own[q] = w < BB_TW && cu < ni && a.p.bb_colhas[col[q]] != 0;
    }
    const bool own_any = own[0] || own[1] || own[2] || own[3];
    const float cs = a.sf;
    float ck[4];                                              /* max sum |wx| of the window's target columns */
#pragma unroll
    for (int q = 0; q < 4; q++) ck[q] = a.p.bb_colk[col[q]];
    if (COARSE) ck[0] = fmaxf(fmaxf(ck[0], ck[1]), fmaxf(ck[2], ck[3]));
    auto load_row = [&](int j) -> float4 {
        const float *zr = zf + (size_t)j * ni;
        float4 v;
        if (VEC) v = *(const float4 *)(zr + col[0]);           /* ni % 4 == 0: the four columns never straddle the seam */
        else { v.x = zr[col[0]]; v.y = zr[col[1]]; v.z = zr[col[2]]; v.w = zr[col[3]]; }
        return v;
    };
    const int jend = j0b + NT - 1;                            /* source rows [j0a, jend) */
    constexpr int G = 4;                                      /* rows per group: loads of the next group are in flight while this one is reduced */
    constexpr int NW = COARSE ? 1 : 4;
    int hpn[NW], hpx[NW], p1n[NW], p1x[NW], p2n[NW], p2x[NW];      /* row r - 1; pair (r - 2, r - 1); pair (r - 3, r - 2) */
#pragma unroll
    for (int q = 0; q < NW; q++) hpn[q] = hpx[q] = p1n[q] = p1x[q] = p2n[q] = p2x[q] = 0;
    float4 nxt[G];
#pragma unroll
    for (int u = 0; u < G; u++) nxt[u] = load_row(min(j0a + u, jend - 1));
    for (int j = j0a; j < jend; j += G) {
        float4 cur[G];
#pragma unroll
        for (int u = 0; u < G; u++) cur[u] = nxt[u];
#pragma unroll
        for (int u = 0; u < G; u++) nxt[u] = load_row(min(j + G + u, jend - 1));        /* (past the tile's last row: a row that is in flight anyway) */
#pragma unroll
        for (int u = 0; u < G; u++) {
            const int jj = j + u;
            if (jj >= jend) break;                            /* wave-uniform */
            const int e0 = bb_key(cur[u].x), e1 = bb_key(cur[u].y), e2 = bb_key(cur[u].z), e3 = bb_key(cur[u].w);
            int hn[NW], hx[NW];
            const int n01 = min(e0, e1), n23 = min(e2, e3), x01 = max(e0, e1), x23 = max(e2, e3);
            kmn = min(kmn, min(n01, n23)); kmx = max(kmx, max(x01, x23));
            if (NT == 1) {
                if (COARSE) { hn[0] = min(n01, n23); hx[0] = max(x01, x23); }
                else { hn[0] = hx[0] = e0; hn[1 % NW] = hx[1 % NW] = e1; hn[2 % NW] = hx[2 % NW] = e2; hn[3 % NW] = hx[3 % NW] = e3; }
            } else {
                const int e4 = __shfl_down(e0, 1, 64);
                if (NT == 2) {
                    if (COARSE) { hn[0] = min(min(n01, n23), e4); hx[0] = max(max(x01, x23), e4); }
                    else { hn[0] = n01; hn[1 % NW] = min(e1, e2); hn[2 % NW] = n23; hn[3 % NW] = min(e3, e4); hx[0] = x01; hx[1 % NW] = max(e1, e2); hx[2 % NW] = x23; hx[3 % NW] = max(e3, e4); }
                } else {
                    const int e5 = __shfl_down(e1, 1, 64), e6 = __shfl_down(e2, 1, 64);
                    if (COARSE) { hn[0] = min(min(n01, n23), min(min(e4, e5), e6)); hx[0] = max(max(x01, x23), max(max(e4, e5), e6)); }
                    else {
                        const int n12 = min(e1, e2), n34 = min(e3, e4), n45 = min(e4, e5), n56 = min(e5, e6);
                        const int x12 = max(e1, e2), x34 = max(e3, e4), x45 = max(e4, e5), x56 = max(e5, e6);
                        hn[0] = min(n01, n23); hn[1 % NW] = min(n12, n34); hn[2 % NW] = min(n23, n45); hn[3 % NW] = min(n34, n56);
                        hx[0] = max(x01, x23); hx[1 % NW] = max(x12, x34); hx[2 % NW] = max(x23, x45); hx[3 % NW] = max(x34, x56);
                    }
                }
            }
            int wn[NW], wx[NW];
#pragma unroll
            for (int q = 0; q < NW; q++) {
                if (NT == 1) { wn[q] = hn[q]; wx[q] = hx[q]; }
                else {
                    const int pn = min(hpn[q], hn[q]), px = max(hpx[q], hx[q]);      /* pair (r - 1, r) */
                    if (NT == 2) { wn[q] = pn; wx[q] = px; }
                    else { wn[q] = min(p2n[q], pn); wx[q] = max(p2x[q], px); p2n[q] = p1n[q]; p2x[q] = p1x[q]; p1n[q] = pn; p1x[q] = px; }
                    hpn[q] = hn[q]; hpx[q] = hx[q];
                }
            }
            const int j0 = jj - NT + 1;
            if (j0 < j0a) continue;                           /* fewer than NT rows so far (wave-uniform) */
            if (!a.p.bb_rowhas[j0]) continue;                 /* no target row starts here (wave-uniform) */
            const float rk = a.p.bb_rowk[j0];                 /* max sum |wy| of the window's target rows (wave-uniform) */
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const float ca = fmaf(fmaf(ck[q], rk, -1.0f), 0.500005f, 1.0e-6f);      /* (K - 1) / 2, rounded up with room for the REAL roundings below */
                const float fx = bb_unkey(wx[q]), fn = bb_unkey(wn[q]);
                const float zabs = __uint_as_float(max(__float_as_uint(fx) & 0x7fffffffu, __float_as_uint(fn) & 0x7fffffffu));
                /* a window of ONE value c interpolates to c exactly (sum w = 1 to REAL*8 rounding, the result is rounded to REAL): upper = lower end */
                const float slack = (a.exact_ok && fx == fn) ? 0.f : fmaf(ca, fx - fn, cs * zabs);
                visit(COARSE ? own_any : own[q], col[q], j0, fx + slack, fn - slack);
            }
        }
    }
    *kmin_out = kmn; *kmax_out = kmx;
}

template <int NT, bool VEC>
__global__ __launch_bounds__(256) void k_bb_bounds(bb_args a)
{
    const int f = blockIdx.z, tx = blockIdx.x, ty = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));      /* wave-uniform: scalar loop control */
    if (ty >= a.nty) return;
    const float *zf = a.zin + (size_t)f * a.in_stride;
    float tub = -INFINITY, tlb = INFINITY, L = -INFINITY, U = INFINITY, cmx = -INFINITY, cmn = INFINITY;
    int kmn, kmx;
    bb_sweep<NT, VEC, true>(a, zf, tx, ty * BB_TH, ty * BB_TH + BB_TH, &kmn, &kmx, [&](bool valid, int, int, float ub, float lb) {
        const bool open = valid && ub != lb;                  /* a window whose values are still to be evaluated; else (ub == lb) its value is known: cmx / cmn */
        tub = fmaxf(tub, open ? ub : -INFINITY); tlb = fminf(tlb, open ? lb : INFINITY);       /* selects, not branches */
        L = fmaxf(L, valid ? lb : -INFINITY); U = fminf(U, valid ? ub : INFINITY);
        cmx = fmaxf(cmx, (valid && !open) ? ub : -INFINITY); cmn = fminf(cmn, (valid && !open) ? lb : INFINITY);
    });
    for (int off = 32; off > 0; off >>= 1) {
        tub = fmaxf(tub, __shfl_down(tub, off, 64)); tlb = fminf(tlb, __shfl_down(tlb, off, 64));
        L = fmaxf(L, __shfl_down(L, off, 64)); U = fminf(U, __shfl_down(U, off, 64));
        cmx = fmaxf(cmx, __shfl_down(cmx, off, 64)); cmn = fminf(cmn, __shfl_down(cmn, off, 64));
    }
    const bool bad = __ballot(kmx >= BB_KEY_POS_INF || kmn <= ~BB_KEY_POS_INF) != 0;
    if ((threadIdx.x & 63) == 0) {
        const size_t t = (size_t)f * a.ntx * a.nty + (size_t)ty * a.ntx + tx;
        /* a tile with a non-finite value: every window of it is evaluated, and nothing of it enters L / U (the interpolating pass's fminf / fmaxf skip
         * NaN results: a bound from a window next to one says nothing about what that pass finds) */
        float *tl = a.tile + BB_TILE_F * t;
        tl[0] = tub; tl[1] = tlb; tl[2] = bad ? -INFINITY : L; tl[3] = bad ? INFINITY : U; tl[4] = bad ? -INFINITY : cmx; tl[5] = bad ? INFINITY : cmn;
        a.tile_bad[t] = bad ? 1u : 0u;
    }
}

__device__ __forceinline__ bool bb_tile_qualifies(const bb_args &a, size_t t, float L, float U)
{
    return a.force_all || a.tile_bad[t] || !(a.tile[BB_TILE_F * t] < L) || !(a.tile[BB_TILE_F * t + 1] > U);
}

__global__ __launch_bounds__(256) void k_bb_reduce(bb_args a)
{
    const int f = blockIdx.x, nt = a.ntx * a.nty;
    const size_t t0 = (size_t)f * nt;
    __shared__ float shL[4], shU[4];
    __shared__ int shc[4];
    __shared__ float shC[4], shD[4];
    float L = -INFINITY, U = INFINITY, C = -INFINITY, D = INFINITY;
    for (int t = threadIdx.x; t < nt; t += 256) {
        const float *tl = a.tile + BB_TILE_F * (t0 + t);
        L = fmaxf(L, tl[2]); U = fminf(U, tl[3]); C = fmaxf(C, tl[4]); D = fminf(D, tl[5]);
    }
    for (int off = 32; off > 0; off >>= 1) {
        L = fmaxf(L, __shfl_down(L, off, 64)); U = fminf(U, __shfl_down(U, off, 64));
        C = fmaxf(C, __shfl_down(C, off, 64)); D = fminf(D, __shfl_down(D, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { shL[threadIdx.x >> 6] = L; shU[threadIdx.x >> 6] = U; shC[threadIdx.x >> 6] = C; shD[threadIdx.x >> 6] = D; }
    __syncthreads();
    L = fmaxf(fmaxf(shL[0], shL[1]), fmaxf(shL[2], shL[3])); U = fminf(fminf(shU[0], shU[1]), fminf(shU[2], shU[3]));
    C = fmaxf(fmaxf(shC[0], shC[1]), fmaxf(shC[2], shC[3])); D = fminf(fminf(shD[0], shD[1]), fminf(shD[2], shD[3]));
    /* the tiles that can still hold an extremum -> the field's tile list (39 744 blocks that look at one tile each and leave cost 3 us per field) */
    __shared__ unsigned s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < nt; t += 256)
        if (bb_tile_qualifies(a, t0 + t, L, U)) a.tlist[t0 + atomicAdd(&s_n, 1u)] = (unsigned)t;
    __syncthreads();
    (void)shc;
    if (threadIdx.x == 0) {
        a.LU[2 * f] = L; a.LU[2 * f + 1] = U;
        a.flags[f] = 0;
        a.ntl[f] = s_n;
        a.count[f] = 0u;
        unsigned *k = a.keys + (size_t)f * a.key_stride;
        k[0] = D <= C ? f2key(D) : 0xFFFFFFFFu; k[1] = D <= C ? f2key(C) : 0u; k[2] = 0u;      /* the values already known: windows of one value */
    }
}

__device__ __forceinline__ void bb_publish(const bb_args &a, int f, float vmin, float vmax)
{
    for (int off = 32; off > 0; off >>= 1) { vmin = fminf(vmin, __shfl_down(vmin, off, 64)); vmax = fmaxf(vmax, __shfl_down(vmax, off, 64)); }
    if ((threadIdx.x & 63) == 0 && vmin <= vmax) {           /* something was evaluated (NaN results never enter: fminf / fmaxf, like the interpolating pass) */
        unsigned *k = a.keys + (size_t)f * a.key_stride;
        atomicMin(&k[0], f2key(vmin)); atomicMax(&k[1], f2key(vmax));
    }
}

/* the tiles that can hold an extremum are swept again: their windows that reach L (or U) are collected in LDS (one wave per tile: a running count in a
 * scalar register, ranks by ballot, no atomics) and appended to the field's list with ONE returning atomic per tile (one per window row and lane window
 * -- 128 per tile on one word per field -- ran at the ~88 atomics per us of a single address: 19 us per cfg2 field) */
template <int NT, bool VEC>
__global__ __launch_bounds__(64) void k_bb_select(bb_args a)
{
    __shared__ unsigned buf[BB_TW * BB_SUB];
    const int f = blockIdx.y;
    if (a.flags[f]) return;
    const int nt = a.ntx * a.nty;
    const float L = a.LU[2 * f], U = a.LU[2 * f + 1];
    const float *zf = a.zin + (size_t)f * a.in_stride;
    const int lane = threadIdx.x & 63;
    const unsigned ntl = a.ntl[f];
  /* a tile goes to BB_TH / BB_SUB blocks, BB_SUB window rows each: one wave sweeping the 32 rows of a tile alone was the whole pass's duration */
  for (unsigned kt = blockIdx.x; kt < ntl * (BB_TH / BB_SUB); kt += gridDim.x) {
    const int t = (int)a.tlist[(size_t)f * nt + kt / (BB_TH / BB_SUB)], sub = (int)(kt % (BB_TH / BB_SUB));
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const bool all = a.force_all != 0 || a.tile_bad[(size_t)f * nt + t] != 0;
    __syncthreads();                                          /* the previous tile's copy-out is done with buf */
    unsigned nloc = 0;                                        /* wave-uniform */
    float vmin = INFINITY, vmax = -INFINITY;
    int kmn, kmx;
    bb_sweep<NT, VEC, false>(a, zf, tx, ty * BB_TH + sub * BB_SUB, ty * BB_TH + sub * BB_SUB + BB_SUB, &kmn, &kmx, [&](bool valid, int i0, int j0, float ub, float lb) {
        const bool known = !all && ub == lb;                   /* a window of one value: nothing to evaluate */
        if (valid && known) { vmin = fminf(vmin, lb); vmax = fmaxf(vmax, ub); }
        const bool cand = valid && !known && (all || !(ub < L) || !(lb > U));
        const unsigned long long m = __ballot(cand);
        if (cand) buf[nloc + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (unsigned)i0 | (unsigned)j0 << 16;
        nloc += (unsigned)__builtin_popcountll(m);
    });
    bb_publish(a, f, vmin, vmax);
    if (!nloc) continue;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&a.count[f], nloc);
    base = (unsigned)__shfl((int)base, 0, 64);
    __syncthreads();
    unsigned *list = a.list + (size_t)f * BB_LIST_CAP;
    for (unsigned k = lane; k < nloc; k += 64) if (base + k < BB_LIST_CAP) list[base + k] = buf[k];
  }
}

/* the value k_sep / k_sepx store at target column c, main row r: the same tables, the same operation order */
template <int DEG>
__device__ __forceinline__ float bb_exact(const ezhip_sep_plan &p, const float *zf, int c, int r)
{
    const ColTaps t = load_col(p.cidx, p.cw, p.ni_dst, c);
    const int nis = p.ni_src, njd = p.nj_dst;
    const float *z0 = zf + (size_t)p.rbase[r] * nis;
    double val;
    if (DEG == 0) val = xpass<0>(z0, t);
    else if (DEG == 1) { const double t0 = xpass<1>(z0, t), t1 = xpass<1>(z0 + nis, t); val = t0 + (t1 - t0) * p.rw[r]; }
    else {
        const double t0 = xpass<3>(z0, t), t1 = xpass<3>(z0 + nis, t), t2 = xpass<3>(z0 + 2 * (size_t)nis, t), t3 = xpass<3>(z0 + 3 * (size_t)nis, t);
        val = fma(p.rw[3 * njd + r], t3, fma(p.rw[2 * njd + r], t2, fma(p.rw[njd + r], t1, p.rw[r] * t0)));
    }
    return (float)val;
}

/* the listed windows: every target point whose stencil is the window, evaluated exactly */
template <int DEG>
__global__ __launch_bounds__(256) void k_bb_eval(bb_args a)
{
    const int f = blockIdx.y;
    if (a.flags[f]) return;
    const unsigned n = a.count[f];
    if (n > (unsigned)a.list_cap) { if (blockIdx.x == 0 && threadIdx.x == 0) a.flags[f] = 1; return; }      /* too many windows can hold the extremum: the field is handed back (every block of the field sees the same count and leaves) */
    const float *zf = a.zin + (size_t)f * a.in_stride;
    const unsigned *list = a.list + (size_t)f * BB_LIST_CAP;
    float vmin = INFINITY, vmax = -INFINITY;
    for (unsigned w = blockIdx.x * 256u + threadIdx.x; w < n; w += gridDim.x * 256u) {
        const unsigned e = list[w];
        const int i0 = (int)(e & 0xFFFFu), j0 = (int)(e >> 16);
        const int c0 = a.p.bb_colstart[i0], c1 = a.p.bb_colstart[i0 + 1], r0 = a.p.bb_rowstart[j0], r1 = a.p.bb_rowstart[j0 + 1];
        for (int ri = r0; ri < r1; ri++) {
            const int r = a.p.bb_rowlist[ri];
            for (int ci = c0; ci < c1; ci++) {
                const float v = bb_exact<DEG>(a.p, zf, a.p.bb_collist[ci], r);
                vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
            }
        }
    }
    bb_publish(a, f, vmin, vmax);
}

template <int DEG>
__global__ __launch_bounds__(SEP_BLOCK) void k_bb_special(bb_args a)
{
    const int f = blockIdx.z;
    if (a.flags[f]) return;
    ezhip_sep_plan &p = a.p;
    p.polevals = a.poles ? a.poles + 2 * f : nullptr;
    const float *zf = a.zin + (size_t)f * a.in_stride;
    const int c = blockIdx.x * SEP_BLOCK + threadIdx.x, cc = min(c, p.ni_dst - 1);
    const float v = sep_special<DEG, 2>(p, nullptr, zf, blockIdx.y, c, cc, c < p.ni_dst, 0.0f);
    bb_publish(a, f, v, v);
}

static void bb_geometry(const ezhip_sep_plan *plan, int *ntx, int *nty)
{
    *ntx = (plan->ni_src + BB_TW - 1) / BB_TW;
    const int nwr = plan->nj_src - plan->bb_ntap + 1;
    *nty = (nwr + BB_TH - 1) / BB_TH;
}
extern "C" size_t ezhip_bb_work_bytes(const ezhip_sep_plan *plan, int nfields)
{
    int ntx, nty; bb_geometry(plan, &ntx, &nty);
    const size_t nt = (size_t)ntx * nty * (size_t)nfields;
    return 4 * BB_TILE_F * nt + 4 * nt + 4 * nt + 8 * (size_t)nfields + 8 * (size_t)nfields + 4 * (size_t)BB_LIST_CAP * (size_t)nfields + 256;
}
static void bb_fill(bb_args *a, const ezhip_sep_plan *plan, const float *d_zin, size_t in_stride, int nfields, unsigned *d_partials, size_t stride_words,
                    int *d_flags, const float *d_poles, void *d_work)
{
    memset(a, 0, sizeof(*a));
    a->p = *plan; a->p.pole_timeout = 0; a->p.vector_mode = 0;
    a->zin = d_zin; a->in_stride = in_stride; a->nfields = nfields;
    bb_geometry(plan, &a->ntx, &a->nty);
    /* the bounds are formed in REAL: the factors are rounded up with room for the (at most four) REAL roundings of slack, upper and lower end */
    a->sf = (float)(plan->bb_s + 4.0e-7);
    const size_t nt = (size_t)a->ntx * a->nty * (size_t)nfields;
    char *w = (char *)d_work;
    a->tile = (float *)w; w += 4 * BB_TILE_F * nt;
    a->tile_bad = (unsigned *)w; w += 4 * nt;
    a->tlist = (unsigned *)w; w += 4 * nt;
    a->LU = (float *)w; w += 8 * (size_t)nfields;
    a->count = (unsigned *)w; w += 4 * (size_t)nfields;
    a->ntl = (unsigned *)w; w += 4 * (size_t)nfields;
    a->list = (unsigned *)w;
    a->keys = d_partials; a->key_stride = stride_words; a->flags = d_flags; a->poles = d_poles;
    /* when is a field handed back (flags)?  When more than an eighth of its windows (at most BB_LIST_CAP) can hold an extremum: evaluating them point by
     * point then costs more than the interpolating pass.  (The tile count alone says little: the second sweep of a tile costs what the first did.) */
    a->max_cand = a->ntx * a->nty;
    const long long nwin = (long long)plan->ni_src * (plan->nj_src - plan->bb_ntap + 1);
    long long cap = nwin / 8;
    if (cap < 8192) cap = 8192;
    if (cap > BB_LIST_CAP) cap = BB_LIST_CAP;
    a->force_all = getenv("EZHIP_BB_FORCE_ALL") ? 1 : 0;      /* tests: every window is evaluated */
    a->list_cap = a->force_all ? BB_LIST_CAP : (int)cap;
    a->exact_ok = plan->bb_s < 1.0e-9 ? 1 : 0;               /* sum w = 1 to REAL*8 rounding: a window of one value c gives c */
}
/* stage 0: everything; 1: the bounds sweep and the reduction (which also initialises the fields' keys and flags); 2: the second sweep of the listed tiles and the evaluation
 * of their windows -- the special rows (ezhip_minmax_bb_special) need stage 1 only and can run beside stage 2 on another stream */
extern "C" int ezhip_minmax_bb_stage(const ezhip_sep_plan *plan, const float *d_zin, size_t in_stride, int nfields, unsigned *d_partials, size_t stride_words,
                                     int *d_flags, const float *d_poles, void *d_work, int stage)
{
    if (!plan->bb_ok || nfields < 1 || plan->ni_src > 65535 || plan->nj_src > 65535) return -2;
    bb_args a;
    bb_fill(&a, plan, d_zin, in_stride, nfields, d_partials, stride_words, d_flags, d_poles, d_work);
    const bool vec = plan->ni_src % 4 == 0 && in_stride % 4 == 0 && ((uintptr_t)d_zin & 15) == 0;
    const dim3 g1((unsigned)a.ntx, (unsigned)((a.nty + 3) / 4), (unsigned)nfields), g2((unsigned)(a.ntx * a.nty * (BB_TH / BB_SUB) < 128 ? a.ntx * a.nty * (BB_TH / BB_SUB) : 128), (unsigned)nfields), g3(64, (unsigned)nfields);
#define BB_LAUNCH(NT, DEG) do { \
        if (stage != 2) { \
            if (vec) hipLaunchKernelGGL((k_bb_bounds<NT, true>), g1, dim3(256), 0, g_stream, a); else hipLaunchKernelGGL((k_bb_bounds<NT, false>), g1, dim3(256), 0, g_stream, a); \
            hipLaunchKernelGGL(k_bb_reduce, dim3((unsigned)nfields), dim3(256), 0, g_stream, a); } \
        if (stage != 1) { \
            if (vec) hipLaunchKernelGGL((k_bb_select<NT, true>), g2, dim3(64), 0, g_stream, a); else hipLaunchKernelGGL((k_bb_select<NT, false>), g2, dim3(64), 0, g_stream, a); \
            hipLaunchKernelGGL(k_bb_eval<DEG>, g3, dim3(256), 0, g_stream, a); } } while (0)
    if (plan->bb_ntap == 4) BB_LAUNCH(4, 3); else if (plan->bb_ntap == 2) BB_LAUNCH(2, 1); else BB_LAUNCH(1, 0);
#undef BB_LAUNCH
    return LAUNCH_CHECK("k_bb");
}
extern "C" int ezhip_minmax_bb(const ezhip_sep_plan *plan, const float *d_zin, size_t in_stride, int nfields, unsigned *d_partials, size_t stride_words,
                               int *d_flags, const float *d_poles, void *d_work)
{
    return ezhip_minmax_bb_stage(plan, d_zin, in_stride, nfields, d_partials, stride_words, d_flags, d_poles, d_work, 0);
}
/* the special (polar) rows of the plan, after (stage 1 of) ezhip_minmax_bb on the same arguments */
extern "C" int ezhip_minmax_bb_special(const ezhip_sep_plan *plan, const float *d_zin, size_t in_stride, int nfields, unsigned *d_partials, size_t stride_words,
                                       int *d_flags, const float *d_poles, void *d_work)
{
    if (!plan->bb_ok) return -2;
    if (plan->n_special <= 0) return 0;
    bb_args a;
    bb_fill(&a, plan, d_zin, in_stride, nfields, d_partials, stride_words, d_flags, d_poles, d_work);
    const dim3 g((unsigned)((plan->ni_dst + SEP_BLOCK - 1) / SEP_BLOCK), (unsigned)plan->n_special, (unsigned)nfields);
    if (plan->degree == 0) hipLaunchKernelGGL(k_bb_special<0>, g, dim3(SEP_BLOCK), 0, g_stream, a);
    else if (plan->degree == 1) hipLaunchKernelGGL(k_bb_special<1>, g, dim3(SEP_BLOCK), 0, g_stream, a);
    else hipLaunchKernelGGL(k_bb_special<3>, g, dim3(SEP_BLOCK), 0, g_stream, a);
    return LAUNCH_CHECK("k_bb_special");
}

/* ===================================================================================== */
/* k_polar_wind : synthetic polar wind rows (vector mode)                                     */
/* ===================================================================================== */
/* ez_calcnpolarwind.c:28-138 / ez_calcspolarwind.c on the device: from the last / first source row of (u,v) to a row of
 * pole winds, through speed/direction, a polar-stereographic frame (ez_llwfgdw / ez_gdwfllw 'N' / 'S' with xg4 from
 * cxgaig/cigaxg), the (sequential REAL) pole value of each component, and back.  blockIdx.x: 0 north, 1 south;
 * One block of 256 threads per row; `plon` = longitudes of the source row (host, once per
 * grid).  out = [u_n, u_s, v_n, v_s] (ni each).  The REAL sinf / cosf / asinf / atan2f of the wind chain proper (k_wind_rotate, d_rotate*: what the grid pair's wind
 * matrix is made from) are libm_exact.h's (GNU libc's, operation by operation: the reference reaches them through the Fortran intrinsics) since round 6: the device
 * library's own differ by <= 2 ulp, which the rotated frame's atan2f(q1, q0) next to a rotated pole (|q| ~ 1e-3) turns into 3e-5 |V| of wind direction -- found by
 * comparing ALL of cfg3's 16 M values with the reference run (tests/test_gpu_wind_pin.py); the sampled rows / columns of the earlier test had missed the 122 points. */
/* (the pole rows of the pair kernels keep the device library's REAL functions: nothing amplifies their <= 2 ulp here -- EZHIP_POLAR_WIND_HOST=1, the C library's own, moves no cfg3
 * value by more than 2e-7 |V| -- and libm_exact.h's versions in the two producer blocks of k_uvt made them the launch's long pole: 69 -> 73 us per cfg3 pair, profiles/r06_experiments.txt.
 * EXACT = true: the standalone k_polar_wind of the exact-winds mode, ezhip_set_wind_exact) */
/* fmodf(x, 360.0f), AMOD(x, 360.) of ez_llwfgdw.inc: x - 360 trunc(x / 360), always exact.  The wind directions that reach it lie in [0, 720) but for polar-stereographic
 * frames with large dgrw: there one subtraction (exact: x and 360 are within a factor of two of each other) replaces the library's remainder loop */
__device__ __forceinline__ float d_fmod360(float x)
{
    if (x >= 0.0f && x < 720.0f) return x >= 360.0f ? x - 360.0f : x;
    return fmodf(x, 360.0f);
}
template <bool EXACT>
__device__ __forceinline__ void d_llwfgdw1(float &z1, float &z2, float xlon, char t, float xg4)
{
    const float RDTODG = 57.295779513082f;
    float uu = z1, vv = z2, spd = sqrtf(uu * uu + vv * vv), dir;
    const float at = (spd == 0.0f || uu == 0.0f) ? 0.0f : (EXACT ? glx_atan2f(vv, uu) : atan2f(vv, uu));
    if (spd == 0.0f) dir = 0.0f;
    else if (t == 'N') dir = (uu == 0.0f) ? ((vv >= 0.0f) ? xlon + xg4 - 90.0f : xlon + xg4 + 90.0f) : xlon + xg4 - RDTODG * at;
    else if (t == 'S') dir = (uu == 0.0f) ? ((vv >= 0.0f) ? 90.0f - xlon + xg4 : 270.0f - xlon + xg4) : 180.0f - xlon + xg4 - RDTODG * at;
    else dir = (uu == 0.0f) ? ((vv >= 0.0f) ? 180.0f : 0.0f) : 270.0f - RDTODG * at;
    dir = d_fmod360(d_fmod360(dir) + 360.0f);
    z1 = spd; z2 = dir;
}
template <bool EXACT>
__device__ __forceinline__ void d_gdwfllw1(float &z1, float &z2, float xlon, char t, float xg4)
{
    const float DGTORD = 1.7453292519943e-2f;
    float psi = t == 'N' ? xlon + xg4 - z2 : t == 'S' ? 180.0f - xlon + xg4 - z2 : 270.0f - z2;
    float u = (EXACT ? glx_cosf(psi * DGTORD) : cosf(psi * DGTORD)) * z1, v = (EXACT ? glx_sinf(psi * DGTORD) : sinf(psi * DGTORD)) * z1;
    z1 = u; z2 = v;
}
template <int CHUNK, bool EXACT = false, int NT = 256>      /* NT: threads of the block that runs it (the standalone launch: 1024 -- its exact trig is the long part) */
__device__ __forceinline__ void polar_wind_body(const int north, float *out, const float *uu, const float *vv, const float *plon2 /* [north row | south row] */,
                                                int ni, int nj, float xg4_n, float xg4_s, int weighted, const float *ax, float *lds /* CHUNK + 4 floats, 16-byte aligned */)
{
    const char hs = north ? 'N' : 'S';
    const float xg4 = north ? xg4_n : xg4_s;
    const float *urow = uu + (north ? (size_t)(nj - 1) * ni : 0), *vrow = vv + (north ? (size_t)(nj - 1) * ni : 0);
    const float *plon = plon2 + (north ? 0 : ni);
    float *pu = out + (north ? 0 : ni), *pv = out + 2 * (size_t)ni + (north ? 0 : ni);
    for (int i = threadIdx.x; i < ni; i += NT) {          /* speed / direction on the lat-lon frame, then polar-stereographic components */
        float a = urow[i], b = vrow[i];
        d_llwfgdw1<EXACT>(a, b, plon[i], 'A', 0.f);
        d_gdwfllw1<EXACT>(a, b, plon[i], hs, xg4);
        pu[i] = a; pv[i] = b;
    }
    __threadfence_block();
    __syncthreads();
    float s0, w0;
    block_poleval2(pu, pv, ni, weighted, ax, lds, CHUNK / 2, s0, w0, NT);
    d_llwfgdw1<EXACT>(s0, w0, 0.0f, hs, xg4);
    __syncthreads();
    for (int i = threadIdx.x; i < ni; i += NT) {
        float spd = s0, wd = (i == 0 || north) ? w0 + plon[i] : w0 - plon[i];
        d_gdwfllw1<EXACT>(spd, wd, plon[i], 'A', 0.f);
        pu[i] = spd; pv[i] = wd;
    }
}
template <bool EXACT>
__global__ __launch_bounds__(1024) void k_polar_wind(float *out, const float *uu, const float *vv, const float *plon2, int ni, int nj, float xg4_n, float xg4_s,
                                                     int weighted, const float *ax)
{
    __shared__ __attribute__((aligned(16))) float lds[POLE_CHUNK + 4];
    polar_wind_body<POLE_CHUNK, EXACT, 1024>(blockIdx.x == 0, out, uu, vv, plon2, ni, nj, xg4_n, xg4_s, weighted, ax, lds);
}
/* ez_corrbgd.inc:20-55 (called at the end of ez_corrval for a Z- or #-on-E source and a 'B' target, ez_corrval.c:146-148): the rows of the
 * target at the poles become their mean -- a sequential REAL sum over the row divided by ni * 1.0 (block_poleval, unweighted).
 * blockIdx.x: 0 = row 1, 1 = row nj; rows: bit 0 = do row 1, bit 1 = do row nj */
__global__ __launch_bounds__(256) void k_corrbgd(float *zout, int ni, int nj, int rows)
{
    __shared__ __attribute__((aligned(16))) float lds[POLE_CHUNK + 4];
    if (!((rows >> blockIdx.x) & 1)) return;
    float *row = zout + (blockIdx.x ? (size_t)(nj - 1) * ni : 0);
    const float m = block_poleval(row, ni, 0, nullptr, lds, POLE_CHUNK);
    for (int i = threadIdx.x; i < ni; i += 256) row[i] = m;
}
extern "C" int ezhip_corrbgd(float *d_zout, int ni, int nj, int hem)
{
    const int rows = ((hem == 0 || hem == 2) ? 1 : 0) | ((hem == 0 || hem == 1) ? 2 : 0);
    if (!rows) return 0;
    hipLaunchKernelGGL(k_corrbgd, dim3(2), dim3(256), 0, g_stream, d_zout, ni, nj, rows);
    return LAUNCH_CHECK("k_corrbgd");
}
extern "C" int ezhip_polar_wind(float *d_out4, const float *d_uu, const float *d_vv, const float *d_plon2, int ni, int nj,
                                float xg4_n, float xg4_s, int weighted, const float *d_ax, int exact)
{
    if (exact) hipLaunchKernelGGL(k_polar_wind<true>, dim3(2), dim3(1024), 0, g_stream, d_out4, d_uu, d_vv, d_plon2, ni, nj, xg4_n, xg4_s, weighted, d_ax);
    else hipLaunchKernelGGL(k_polar_wind<false>, dim3(2), dim3(1024), 0, g_stream, d_out4, d_uu, d_vv, d_plon2, ni, nj, xg4_n, xg4_s, weighted, d_ax);
    return LAUNCH_CHECK("k_polar_wind");
}

/* the wind chain of a grid pair as its 2 x 2 matrix (ezhip_wind_matrix).  The chain itself passes through speed and direction, and its REAL
 * sqrt(u*u + v*v) (ez_llwfgdw.inc:91-165) overflows for |V| > 1.8e19 -- raw cubic extrapolation far outside a polar-stereographic source gets there:
 * the speed is then inf, and what follows it is NaN in both components when the target frame is rotated (ez_uvacart.inc: inf - inf in one of the first
 * two cartesian components for every sign pattern, then mxm) and inf with the sign of the component otherwise (ez_gdwfllw.inc: cos(psi) * inf).
 * The matrix has no square in it; this reproduces the chain's overflow (found by tools/fuzz_vs_ref2.py against the reference build). */
__device__ __forceinline__ void d_wind_matrix_apply(float mx, float my, float mz, float mw, float u, float v, int dst_rot, float &a, float &b)
{
    a = mx * u + my * v; b = mz * u + mw * v;
    const float s2 = dst_rot ? u * u + v * v : a * a + b * b;
    if (!(s2 <= 3.402823466e+38f)) {
        const float inf = __builtin_inff(), nan = __builtin_nanf("");
        if (dst_rot) { a = nan; b = nan; }
        else { a = (a == 0.0f || a != a) ? nan : copysignf(inf, a); b = (b == 0.0f || b != b) ? nan : copysignf(inf, b); }
    }
}
/* the matrix of point o.  half: the set's chain is a pure rotation to rounding at every point (checked when the matrix is built) and the matrix is kept as ONE word
 * per point (round 5; a pair (a, b) before: 8 bytes per point and call, 64 of the 192 MB cfg3's pair kernel streams): the SMALLER of a, b as a REAL whose two lowest
 * mantissa bits say which one it is (bit 0: 1 = a) and the sign of the other (bit 1); the other is +- sqrt(1 - s^2).  |s| <= 0.708, so both come back to ~1.5e-7
 * absolute (the packing rounds s to 22 mantissa bits; the square root of 1 - s^2 is no worse than s) -- the chain's own roundings leave a^2 + b^2 - 1 at a few
 * 1e-7 too.  EVERY consumer of the half form decodes this word (first call, staged tiles, special points, k_wind_apply): one set of values per grid set.
 * The same two unconditional loads in both forms (a conditional one costs a wait at the join): the half form's second load brings the next point's word, ignored;
 * the buffer ends with 16 spare bytes */
typedef float wm_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned rot_pack(float a, float b)
{
    const bool a_small = fabsf(a) < fabsf(b);
    const float sm = a_small ? a : b, other = a_small ? b : a;
    return ((__float_as_uint(sm) + 2u) & ~3u) | (a_small ? 1u : 0u) | (other < 0.0f ? 2u : 0u);
}
__device__ __forceinline__ wm_f2 rot_unpack(unsigned q)
{
    const float sm = __uint_as_float(q & ~3u);
    float other = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - sm * sm));      /* (v_sqrt_f32 itself, 1 ulp: one instruction, the same in every kernel; the IEEE sequence costs ten) */
    other = (q & 2u) ? -other : other;
    return (q & 1u) ? wm_f2{sm, other} : wm_f2{other, sm};
}
__device__ __forceinline__ void wind_m_load(const void *M, int half, size_t o, wm_f2 &lo, wm_f2 &hi)
{
    if (half) {
        const unsigned *mq = (const unsigned *)M + o;
        const unsigned q = __builtin_nontemporal_load(mq), q2 = __builtin_nontemporal_load(mq + 1);
        lo = rot_unpack(q); hi = wm_f2{__uint_as_float(q2), 0.0f};
    } else {
        const wm_f2 *mp = (const wm_f2 *)((const char *)M + (o << 4));
        lo = __builtin_nontemporal_load(mp);
        hi = __builtin_nontemporal_load(mp + 1);
    }
}
__device__ __forceinline__ void wind_m_apply(wm_f2 lo, wm_f2 hi, int half, float u, float v, int dst_rot, float &a, float &b)
{
    if (half) d_wind_matrix_apply(lo.x, lo.y, -lo.y, lo.x, u, v, dst_rot, a, b);       /* (a wave-uniform branch: as selects the two forms cost the pair kernel two spilled registers) */
    else d_wind_matrix_apply(lo.x, lo.y, hi.x, hi.y, u, v, dst_rot, a, b);
}
/* ===================================================================================== */
/* k_pts : generic per-point interpolation (restates the reference leaf kernels)            */
/* ===================================================================================== */

/* Source-field accessor with synthetic pole rows above j2 / below j1 (the 4-row strips of
 * ez_fillnpole.inc / ez_fillspole.inc without materialising them). */
struct FieldAcc {
    const float *z; int ni, j1, j2;
    float pole_n, pole_s;
    const float *prow_n, *prow_s;       /* vector mode: per-column pole rows */
    /* ONE unconditional load per access, the row chosen by selects: behind the two row tests the 16 gathers of a strip point were 16 dependent round trips (the
     * special points' kernels -- a fraction of a percent of the points -- ran 9.5 us behind every cfg3 call: latency, nothing else) */
    __device__ __forceinline__ float operator()(int i, int j) const
    {
        const bool hi = j > j2, lo = j < j1;
        const float *row = hi ? prow_n : (lo ? prow_s : z + (size_t)(j - j1) * ni);
        const float v = (row ? row : z)[i - 1];
        return (hi && !prow_n) ? pole_n : ((lo && !prow_s) ? pole_s : v);
    }
};

/* Main-zone accessor: the leaf kernels clamp j to [j1, j2], so no pole row can be touched and the 16 gathers of a
 * point are unconditional (with FieldAcc every gather sits behind two row tests and the loads of a point serialise:
 * k_pts 392 us per 8 M cfg3 points). */
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));      /* 16-byte load at 4-byte alignment (global_load_dwordx4) */
struct PlainAcc {
    const float *z; int ni, j1;
    __device__ __forceinline__ float operator()(int i, int j) const { return z[(size_t)(j - j1) * ni + (i - 1)]; }
};
/* the four taps of one stencil row: ONE 16-byte load when the columns are consecutive (everywhere but at the longitude
 * seam) -- a gather instruction costs the same for 4 or 16 bytes per lane, and the 16 dword gathers of a bicubic point
 * were most of k_pts' time */
template <class A>
__device__ __forceinline__ void row_taps(const A &Z, int im1, int i, int ip1, int ip2, int jj, double &z1, double &z2, double &z3, double &z4)
{
    z1 = Z(im1, jj); z2 = Z(i, jj); z3 = Z(ip1, jj); z4 = Z(ip2, jj);
}
template <>
__device__ __forceinline__ void row_taps<PlainAcc>(const PlainAcc &Z, int im1, int i, int ip1, int ip2, int jj, double &z1, double &z2, double &z3, double &z4)
{
    if (i == im1 + 1 && ip1 == i + 1 && ip2 == i + 2) {
        const f4u v = *(const f4u *)(Z.z + (size_t)(jj - Z.j1) * Z.ni + (im1 - 1));
        z1 = v.x; z2 = v.y; z3 = v.z; z4 = v.w;
    } else { z1 = Z(im1, jj); z2 = Z(i, jj); z3 = Z(ip1, jj); z4 = Z(ip2, jj); }
}

__device__ __forceinline__ double d_zlin(double a, double b, double t) { return a + (b - a) * t; }
__device__ __forceinline__ double d_cubic(double z1, double z2, double z3, double z4, double dx)
{   /* cubic8.cdk: the two literals are default-REAL constants */
    const double c6 = (double)0.1666666666666f, c3 = (double)0.3333333333333f;
    return ((((z4 - z1) * c6 + 0.5 * (z2 - z3)) * dx + 0.5 * (z1 + z3) - z2) * dx + z3 - c6 * z4 - 0.5 * z2 - c3 * z1) * dx + z2;
}
__device__ __forceinline__ double d_fa(double a1, double a2, double a3, double a4, double x, double x1, double x2, double x3)
{ return a1 + (x - x1) * (a2 + (x - x2) * (a3 + a4 * (x - x3))); }
__device__ __forceinline__ double d_fa2(double c1, double a1, double a2) { return c1 * (a2 - a1); }
__device__ __forceinline__ double d_fa3(double c1, double c2, double c3, double a1, double a2, double a3)
{ return c2 * (c3 * (a3 - a2) - c1 * (a2 - a1)); }
__device__ __forceinline__ double d_fa4(double c1, double c2, double c3, double c4, double c5, double c6, double a1, double a2, double a3, double a4)
{ return c4 * (c5 * (c6 * (a4 - a3) - c3 * (a3 - a2)) - c2 * (c3 * (a3 - a2) - c1 * (a2 - a1))); }

__device__ __forceinline__ int d_nint(float v) { return (int)lroundf(v); }

/* ez_rgdint_0.inc:20-35 */
template <class A> __device__ __forceinline__ float p_rgdint_0(const A &Z, float px, float py, int ni, int j1, int j2)
{
    int i = min(ni, max(1, d_nint(px)));
    int j = min(j2, max(j1, d_nint(py)));
    return Z(i, j);
}
/* ez_rgdint_1_nw.inc:20-44 */
template <class A> __device__ __forceinline__ float p_rgdint_1_nw(const A &Z, float px, float py, int ni, int j1, int j2)
{
    int i = min(ni - 1, max(1, (int)px));
    int j = min(j2 - 1, max(j1, (int)py));
    double dx = (double)(px - (float)i), dy = (double)(py - (float)j);
    double y2 = d_zlin((double)Z(i, j), (double)Z(i + 1, j), dx);
    double y3 = d_zlin((double)Z(i, j + 1), (double)Z(i + 1, j + 1), dx);
    return (float)d_zlin(y2, y3, dy);
}
/* ez_rgdint_1_w.inc:20-51 */
template <class A> __device__ __forceinline__ float p_rgdint_1_w(const A &Z, float px, float py, int ni, int j1, int j2, int wrap)
{
    int limite = ni + 2 - wrap;
    int i = min(ni - 2 + wrap, max(1, (int)px));
    int j = min(j2 - 1, max(j1, (int)py));
    int ip1 = i + 1;
    if (wrap > 0 && (i == (ni - 2 + wrap))) ip1 = (limite + i + 1) % limite;
    double dx = (double)(px - (float)i), dy = (double)(py - (float)j);
    double y2 = d_zlin((double)Z(i, j), (double)Z(ip1, j), dx);
    double y3 = d_zlin((double)Z(i, j + 1), (double)Z(ip1, j + 1), dx);
    return (float)d_zlin(y2, y3, dy);
}
template <class A> __device__ __forceinline__ float cubic_rows(const A &Z, int im1, int i, int ip1, int ip2, int j, double dx, double dy)
{
    double a[4], b[4], c[4], d[4];
    row_taps(Z, im1, i, ip1, ip2, j - 1, a[0], a[1], a[2], a[3]);
    row_taps(Z, im1, i, ip1, ip2, j, b[0], b[1], b[2], b[3]);
    row_taps(Z, im1, i, ip1, ip2, j + 1, c[0], c[1], c[2], c[3]);
    row_taps(Z, im1, i, ip1, ip2, j + 2, d[0], d[1], d[2], d[3]);
    double y1 = d_cubic(a[0], a[1], a[2], a[3], dx);
    double y2 = d_cubic(b[0], b[1], b[2], b[3], dx);
    double y3 = d_cubic(c[0], c[1], c[2], c[3], dx);
    double y4 = d_cubic(d[0], d[1], d[2], d[3], dx);
    return (float)d_cubic(y1, y2, y3, y4, dy);
}
/* ez_rgdint_3_nw.inc:20-77 */
template <class A> __device__ __forceinline__ float p_rgdint_3_nw(const A &Z, float px, float py, int ni, int j1, int j2)
{
    int i = min(ni - 2, max(2, (int)px));
    int j = min(j2 - 2, max(j1 + 1, (int)py));
    return cubic_rows(Z, i - 1, i, i + 1, i + 2, j, (double)(px - (float)i), (double)(py - (float)j));
}
__device__ __forceinline__ void wrap_cols_regular(int ni, int wrap, int limite, int &i, int &im1, int &ip1, int &ip2)
{   /* ez_rgdint_3_w.inc:72-90 (literal, including the wrap==1 remaps) */
    im1 = (limite + i - 1) % limite; ip1 = (limite + i + 1) % limite; ip2 = (limite + i + 2) % limite;
    if (im1 == 0) im1 = ni;
    if (i == 0) i = ni;
    if (ip1 == 0) ip1 = ni;
    if (ip2 == 0) ip2 = ni;
    if (wrap == 1) { if (ip2 == ni) ip2 = 2; if (im1 == ni) im1 = ni - 1; }
}
/* ez_rgdint_3_w.inc:20-108 (nnc = 0), ez_rgdint_3_wnnc.inc:20-107 (nnc = 1) */
template <class A> __device__ __forceinline__ float p_rgdint_3_w(const A &Z, float px, float py, int ni, int j1, int j2, int wrap, int nnc)
{
    int limite = ni + 2 - wrap;
    int i = min(ni - 2 + wrap, max(1, max(2 - wrap, (int)px)));
    int j = min(j2 - 2, max(j1 + 1, (int)py));
    int im1, ip1, ip2;
    bool seam = nnc ? ((wrap > 0 && i <= 1) || i >= (ni - 1)) : (wrap > 0);
    if (seam) wrap_cols_regular(ni, wrap, limite, i, im1, ip1, ip2);
    else { im1 = i - 1; ip1 = i + 1; ip2 = i + 2; }
    return cubic_rows(Z, im1, i, ip1, ip2, j, (double)(px - (float)i), (double)(py - (float)j));
}
/* ez_irgdint_1_nw.inc:20-50 */
template <class A> __device__ __forceinline__ float p_irgdint_1_nw(const A &Z, float px, float py, const float *ax, const float *ay, int ni, int nj)
{
    int i = min(ni - 1, max(1, (int)px));
    int j = min(nj - 1, max(1, (int)py));
    double x1 = ax[i - 1], x2 = ax[i];
    double x = (double)ax[i - 1] + (x2 - x1) * (double)(px - (float)i);
    double y = (double)(ay[j - 1] + (ay[j] - ay[j - 1]) * (py - (float)j));
    double dx = (x - x1) / (x2 - x1);
    double dy = (y - (double)ay[j - 1]) / (double)(ay[j] - ay[j - 1]);
    double y1 = d_zlin((double)Z(i, j), (double)Z(i + 1, j), dx);
    double y2 = d_zlin((double)Z(i, j + 1), (double)Z(i + 1, j + 1), dx);
    return (float)d_zlin(y1, y2, dy);
}
/* ez_irgdint_1_w.inc:20-64 (ay indexed from j1) */
template <class A> __device__ __forceinline__ float p_irgdint_1_w(const A &Z, float px, float py, const float *ax, const float *ay, int ni, int j1, int j2, int wrap)
{
    int limite = ni + 2 - wrap;
    int i = min(ni - 2 + wrap, max(1, (int)px));
    int j = min(j2 - 1, max(j1 + 1, (int)py));
    if (j < 0) j = j - 1;
    int ip1 = i + 1;
    double x1 = ax[i - 1], x2 = 0.0;
    if (ip1 <= ni) x2 = ax[ip1 - 1];
    if (wrap > 0 && (i == (ni - 2 + wrap))) { ip1 = (limite + i + 1) % limite; x2 = (double)(ax[1] + ax[ni - 1]); }
    float ayj = ay[j - j1], ayj1 = ay[j + 1 - j1];
    double x = x1 + (x2 - x1) * (double)(px - (float)i);
    double y = (double)(ayj + (ayj1 - ayj) * (py - (float)j));
    double dx = (x - x1) / (x2 - x1);
    double dy = (y - (double)ayj) / (double)(ayj1 - ayj);
    double y1 = d_zlin((double)Z(i, j), (double)Z(ip1, j), dx);
    double y2 = d_zlin((double)Z(i, j + 1), (double)Z(ip1, j + 1), dx);
    return (float)d_zlin(y1, y2, dy);
}
/* ez_irgdint_3_nw.inc:20-168: the statement functions are REAL there (results rounded to float) */
/* Newton coefficient tables: the reference's layout is cx(ni,6) (6 strided loads per point); the main kernel reads a
 * device copy laid out [index][8] (two aligned 16-byte loads) */
template <bool AOS> __device__ __forceinline__ double coef(const float *c, int k, int idx, int n) { return AOS ? (double)c[idx * 8 + k] : (double)c[k * n + idx]; }

template <class A, bool AOS = false> __device__ __forceinline__ float p_irgdint_3_nw(const A &Z, float px, float py, const float *ax, const float *ay,
                                                   const float *cx, const float *cy, int i1, int i2, int j1, int j2)
{
    const int ni = i2 - i1 + 1, nnj = j2 - j1 + 1;
#define RF(e) ((double)(float)(e))
    int i = min(i2 - 2, max(i1 + 1, (int)px));
    int j = min(j2 - 2, max(j1 + 1, (int)py));
    const float *a = ax - i1, *b = ay - j1;
    double x = (double)(a[i] + (a[i + 1] - a[i]) * (px - (float)i));
    double y = (double)(b[j] + (b[j + 1] - b[j]) * (py - (float)j));
    double x1 = a[i - 1], x2 = a[i], x3 = a[i + 1];
    double y1 = b[j - 1], y2 = b[j], y3 = b[j + 1];
    double c1 = coef<AOS>(cx, 0, i - i1, ni), c2 = coef<AOS>(cx, 1, i - i1, ni), c3 = coef<AOS>(cx, 2, i - i1, ni), c4 = coef<AOS>(cx, 3, i - i1, ni), c5 = coef<AOS>(cx, 4, i - i1, ni), c6 = coef<AOS>(cx, 5, i - i1, ni);
    double bb[4];
    for (int r = 0; r < 4; r++) {
        int jj = j - 1 + r;
        double z1, z2, z3, z4;
        row_taps(Z, i - 1, i, i + 1, i + 2, jj, z1, z2, z3, z4);
        double a2 = RF(d_fa2(c1, z1, z2));
        double a3 = RF(d_fa3(c1, c2, c3, z1, z2, z3));
        double a4 = RF(d_fa4(c1, c2, c3, c4, c5, c6, z1, z2, z3, z4));
        bb[r] = RF(d_fa(z1, a2, a3, a4, x, x1, x2, x3));
    }
    double d1 = coef<AOS>(cy, 0, j - j1, nnj), d2 = coef<AOS>(cy, 1, j - j1, nnj), d3 = coef<AOS>(cy, 2, j - j1, nnj), d4 = coef<AOS>(cy, 3, j - j1, nnj), d5 = coef<AOS>(cy, 4, j - j1, nnj), d6 = coef<AOS>(cy, 5, j - j1, nnj);
    double b12 = RF(d_fa2(d1, bb[0], bb[1]));
    double b13 = RF(d_fa3(d1, d2, d3, bb[0], bb[1], bb[2]));
    double b14 = RF(d_fa4(d1, d2, d3, d4, d5, d6, bb[0], bb[1], bb[2], bb[3]));
    return (float)d_fa(bb[0], b12, b13, b14, y, y1, y2, y3);
#undef RF
}
/* seam logic of ez_irgdint_3_w.inc:100-156 / ez_irgdint_3_wnnc.inc:100-156 */
__device__ __forceinline__ void irr_cols(const float *ax, int ni, int wrap, int i, int ip2_wrap1,
                                         int &im1, int &ip1, int &ip2, float &x1, float &x2, float &x3, float &x4)
{
#define AX(k) ax[(k) - 1]
    im1 = i - 1; ip1 = i + 1; ip2 = i + 2;
    x1 = x2 = x3 = x4 = 0.f;
    if (wrap == 1 && (i <= 1 || i >= (ni - wrap))) {
        if (i == 1) { im1 = ni - 1; ip1 = 2; ip2 = 3; x1 = AX(ni - 1) - 360.0f; x2 = AX(1); x3 = AX(2); x4 = AX(3); }
        if (i == (ni - 1)) { im1 = ni - 2; ip1 = ni; ip2 = ip2_wrap1; x1 = AX(ni - 2); x2 = AX(ni - 1); x3 = AX(ni); x4 = AX(2) + 360.0f; }
    } else if (wrap == 2 && (i <= 1 || i > (ni - wrap))) {
        if (i == 1) { im1 = ni; ip1 = 2; ip2 = 3; x1 = AX(ni) - 360.0f; x2 = AX(1); x3 = AX(2); x4 = AX(3); }
        if (i == (ni - 1)) { im1 = ni - 2; ip1 = ni; ip2 = 1; x1 = AX(ni - 2); x2 = AX(ni - 1); x3 = AX(ni); x4 = AX(1) + 360.0f; }
        if (i == ni) { im1 = ni - 1; ip1 = 1; ip2 = 2; x1 = AX(ni - 1); x2 = AX(ni); x3 = AX(1) + 360.0f; x4 = AX(2) + 360.0f; }
    } else {
        x1 = AX(im1); x2 = AX(i); x3 = AX(ip1); x4 = AX(ip2);
    }
#undef AX
}
/* ez_irgdint_3_w.inc:20-235 */
template <class A, bool AOS = false> __device__ __forceinline__ float p_irgdint_3_w(const A &Z, float px, float py, const float *ax, const float *ay,
                                                  const float *cx, const float *cy, int ni, int j1, int j2, int wrap)
{
    const int nnj = j2 - j1 + 1;
    int i = min(ni - 2 + wrap, max(1, max(2 - wrap, (int)px)));
    int j = min(j2 - 2, max(j1 + 1, (int)py));
    int im1, ip1, ip2; float x1, x2, x3, x4;
    irr_cols(ax, ni, wrap, i, 2, im1, ip1, ip2, x1, x2, x3, x4);
    const float *b = ay - j1;
    double x = (double)(x2 + (x3 - x2) * (px - (float)i));
    double y = (double)(b[j] + (b[j + 1] - b[j]) * (py - (float)j));
    float y1 = b[j - 1], y2 = b[j], y3 = b[j + 1];
    double c1 = coef<AOS>(cx, 0, i - 1, ni), c2 = coef<AOS>(cx, 1, i - 1, ni), c3 = coef<AOS>(cx, 2, i - 1, ni), c4 = coef<AOS>(cx, 3, i - 1, ni), c5 = coef<AOS>(cx, 4, i - 1, ni), c6 = coef<AOS>(cx, 5, i - 1, ni);
    double bb[4];
    for (int r = 0; r < 4; r++) {
        int jj = j - 1 + r;
        double z1, z2, z3, z4;
        row_taps(Z, im1, i, ip1, ip2, jj, z1, z2, z3, z4);
        double a2 = d_fa2(c1, z1, z2);
        double a3 = d_fa3(c1, c2, c3, z1, z2, z3);
        double a4 = d_fa4(c1, c2, c3, c4, c5, c6, z1, z2, z3, z4);
        bb[r] = d_fa(z1, a2, a3, a4, x, (double)x1, (double)x2, (double)x3);
    }
    double d1 = coef<AOS>(cy, 0, j - j1, nnj), d2 = coef<AOS>(cy, 1, j - j1, nnj), d3 = coef<AOS>(cy, 2, j - j1, nnj), d4 = coef<AOS>(cy, 3, j - j1, nnj), d5 = coef<AOS>(cy, 4, j - j1, nnj), d6 = coef<AOS>(cy, 5, j - j1, nnj);
    double b12 = d_fa2(d1, bb[0], bb[1]);
    double b13 = d_fa3(d1, d2, d3, bb[0], bb[1], bb[2]);
    double b14 = d_fa4(d1, d2, d3, d4, d5, d6, bb[0], bb[1], bb[2], bb[3]);
    return (float)d_fa(bb[0], b12, b13, b14, y, (double)y1, (double)y2, (double)y3);
}
/* Both components of a wind at one point, ez_irgdint_3_w.inc:20-235 regrouped: the Newton form is LINEAR in the four values of a row,
 *   fa = z1 + t1 (a2 + t2 (a3 + t3 a4)) = z1 + A (z2 - z1) + B (z3 - z2) + G (z4 - z3),   t_k = x - x_k,
 *   A = t1 c1 (1 - t2 c2 (1 - t3 c4)),  B = t1 t2 c3 (c2 - t3 c4 (c5 + c2)),  G = t1 t2 t3 c4 c5 c6
 * with the reference's own REAL coefficient tables c1..c6 (ez_nwtncof) -- the same polynomial in the same differences, another association of
 * the REAL*8 operations (results differ from the literal form by a few ulp of REAL*8, 1e-16 of the values; winds are compared at 1e-5 |V|).  A, B, G
 * depend on the point only: 4 rows x 2 fields share them, as do the three of the y direction: 190 instead of 375 VALU instructions per point pair
 * in k_pts2, which is bound by them (six waves per SIMD, each 18 % active: the issue slots of the SIMD are taken).  Scalars (k_pts) keep the literal form. */
/* (round 4) ... and once more as WEIGHTS of the four values: z1 + A (z2 - z1) + B (z3 - z2) + G (z4 - z3) = (1 - A) z1 + (A - B) z2 + (B - G) z3 + G z4: one
 * multiplication and three fused multiply-adds per row and field instead of three subtractions and three fma (the pair kernels are bound by their REAL*8-rate
 * VALU work: SQ_ACTIVE_INST_VALU 88 % of the SIMD, profiles/r04_experiments.txt).  The same polynomial, another association once more; every kernel of the
 * pair path (k_pts2_irgd3w, its seam variant, k_uvt) uses this one form, so a grid set's first call and its later calls return the same bits. */
struct NewtonW { double a, b, g, w0, w1, w2; };
__device__ __forceinline__ NewtonW newton_w52(double c1, double c2, double c3, double c4, double c5, double c6, double c52 /* c5 + c2 */, double t1, double t2, double t3)
{
    NewtonW w;
    const double t12 = t1 * t2, t3c4 = t3 * c4;
    w.a = t1 * c1 * (1.0 - t2 * c2 * (1.0 - t3c4));
    w.b = t12 * c3 * (c2 - t3c4 * c52);
    w.g = t12 * t3c4 * c5 * c6;
    w.w0 = 1.0 - w.a; w.w1 = w.a - w.b; w.w2 = w.b - w.g;
    return w;
}
__device__ __forceinline__ NewtonW newton_w(double c1, double c2, double c3, double c4, double c5, double c6, double t1, double t2, double t3)
{
    return newton_w52(c1, c2, c3, c4, c5, c6, c5 + c2, t1, t2, t3);
}
/* (round 5) The 2 x 20 multiply-adds of a wind pair in REAL: the sixteen cells of a component are combined row by row, rows by the y weights, with fused REAL
 * multiply-adds -- as PAIRS (u, v) where the cells lie that way (k_uvt's LDS image: v_pk_fma_f32, 20 packed instructions per point pair instead of 32 conversions
 * + 40 REAL*8-rate operations).  Every kernel of the pair path evaluates a point with pair_eval below, so a set's first call (gathering) and its later calls
 * (staged windows) return the same bits; a packed and a scalar fused multiply-add round identically. */
typedef float pk2 __attribute__((ext_vector_type(2)));
struct PairW { float x0, x1, x2, x3, y0, y1, y2, y3; };
__device__ __forceinline__ pk2 pk_bc(float w) { return pk2{w, w}; }
__device__ __forceinline__ pk2 pair_row(const PairW &w, pk2 z1, pk2 z2, pk2 z3, pk2 z4)
{
    return __builtin_elementwise_fma(z4, pk_bc(w.x3), __builtin_elementwise_fma(z3, pk_bc(w.x2), __builtin_elementwise_fma(z2, pk_bc(w.x1), z1 * pk_bc(w.x0))));
}
__device__ __forceinline__ pk2 pair_cols(const PairW &w, pk2 r0, pk2 r1, pk2 r2, pk2 r3)
{
    return __builtin_elementwise_fma(r3, pk_bc(w.y3), __builtin_elementwise_fma(r2, pk_bc(w.y2), __builtin_elementwise_fma(r1, pk_bc(w.y1), r0 * pk_bc(w.y0))));
}
/* The eight weights in REAL too (away from the longitude seam), in Lagrange's form from the set-up's 32-byte records {x1 .. x4, d1 .. d4}
 * (ezhip_pts_plan.xrec8 / yrec8: d_k = 1 / prod_{m != k} (x_k - x_m) from REAL*8): w_k = d_k prod_{m != k} (x - x_m) -- the interpolating cubic of the four
 * values, the polynomial the reference's Newton form evaluates; 14 REAL operations per direction instead of 21 REAL*8 ones + 4 conversions, records of 32
 * instead of 80 bytes in LDS.  With the REAL sums above: <= 3.9e-7 |V| from the all-REAL*8 evaluation on cfg3's golden rows (bar: 1e-5 |V|). */
typedef float f4a16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lagrange_w(float x, f4a16 xs, f4a16 d, float &w0, float &w1, float &w2, float &w3)
{
    const float t1 = x - xs.x, t2 = x - xs.y, t3 = x - xs.z, t4 = x - xs.w;
    const float p12 = t1 * t2, p34 = t3 * t4;
    w0 = (t2 * p34) * d.x; w1 = (t1 * p34) * d.y; w2 = (p12 * t4) * d.z; w3 = (p12 * t3) * d.w;
}
__device__ __forceinline__ PairW pair_weights_lagrange(float px, float py, int i, int j, f4a16 xs, f4a16 xd, f4a16 ys, f4a16 yd)
{
    const float x = xs.y + (xs.z - xs.y) * (px - (float)i);          /* REAL, as ez_irgdint_3_w.inc:158 - 159 */
    const float y = ys.y + (ys.z - ys.y) * (py - (float)j);
    PairW w;
    lagrange_w(x, xs, xd, w.x0, w.x1, w.x2, w.x3);
    lagrange_w(y, ys, yd, w.y0, w.y1, w.y2, w.y3);
    return w;
}
/* The REAL sums are as good as the stencil is LARGE: their error is ~4e-7 of M = the largest |cell| under the stencil (both components), whatever comes out.
 * Where the wind is nearly calm next to stronger winds (a coarse source, a calm point between two jets) that is more than 1e-5 of |V| -- the reference build found
 * such a point (tools/fuzz_vs_ref2.py 600 7: 3.8e-5 |V| at |V| = 0.037 under cells of +- 15).  And the reference's own polynomial is not the exact interpolating
 * cubic there: its Newton coefficients are REAL-rounded reciprocals, which moves it by ~1e-7 M (exact Lagrange weights in REAL*8 were still 4e-5 |V| off at that
 * point).  So: a point whose larger component is below M / 3 (M / 8 until round 6; M = the largest |cell| of the stencil's two central rows or |w_y| x the largest |cell| of an outer row, whichever is larger) is evaluated once more the reference's way -- its Newton form with its REAL coefficient tables, REAL*8
 * throughout (the pair path's arithmetic of rounds 3 - 4) --, everywhere else the REAL result stands (see PAIR_RULE below for the measured constants).  The test reads the REAL result: deterministic, the same in every kernel of the pair path.  Calm points are isolated: a few waves in
 * a thousand take the second path. */
/* everything by address (its REAL*8 temporaries must not be alive next to the common path's registers: k_uvt runs it as a second pass over the flagged points of a
 * thread, after the last store).  cu / cv: the first cell of the stencil (row j - 1, column i - 1) of either component, cstep floats between columns, cstride
 * between rows; xrec / yrec: the point's 32-byte records (the axis entries); cx8 / cy8: the reference's REAL Newton coefficients of column i / row j ([index][8]).
 * The arithmetic is the pair path's of rounds 3 - 4: ez_irgdint_3_w.inc's Newton form with ITS coefficient tables, regrouped as weights, REAL*8 throughout. */
template <class P>
__device__ __forceinline__ pk2 pair_eval_real8(P cu, P cv, int cstep, int cstride, P xrec, P yrec, const float *cx8, const float *cy8, float px, float py, int i, int j)
{
    const float x1 = xrec[0], x2 = xrec[1], x3 = xrec[2], y1 = yrec[0], y2 = yrec[1], y3 = yrec[2];
    const double x = (double)(x2 + (x3 - x2) * (px - (float)i)), y = (double)(y2 + (y3 - y2) * (py - (float)j));
    const NewtonW wx = newton_w((double)cx8[0], (double)cx8[1], (double)cx8[2], (double)cx8[3], (double)cx8[4], (double)cx8[5], x - (double)x1, x - (double)x2, x - (double)x3);
    const NewtonW wy = newton_w((double)cy8[0], (double)cy8[1], (double)cy8[2], (double)cy8[3], (double)cy8[4], (double)cy8[5], y - (double)y1, y - (double)y2, y - (double)y3);
    const double wr[4] = {wy.w0, wy.w1, wy.w2, wy.g};
    double su = 0.0, sv = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const P ru_ = cu + r * cstride, rv_ = cv + r * cstride;
        su = fma(wr[r], fma(wx.g, (double)ru_[3 * cstep], fma(wx.w2, (double)ru_[2 * cstep], fma(wx.w1, (double)ru_[cstep], wx.w0 * (double)ru_[0]))), su);
        sv = fma(wr[r], fma(wx.g, (double)rv_[3 * cstep], fma(wx.w2, (double)rv_[2 * cstep], fma(wx.w1, (double)rv_[cstep], wx.w0 * (double)rv_[0]))), sv);
    }
    return pk2{(float)su, (float)sv};
}
/* REAL result and "is it trustworthy" (see above): the caller sends the point to pair_eval_real8 when not.
 * PAIR_RULE (round 6: 3, was 8): against the reference build over all of cfg3 and the adversarial fields of tests/test_gpu_wind_pin.py a REAL result is off by at most
 * 2.5e-6 M + 2e-6 |V| (the second term is the reference chain's own noise -- its wind direction passes through REAL degrees -- and stands for the REAL*8 pass too);
 * a point keeps its REAL result only where M <= 3 x its larger component: 3 x 2.5e-6 + 2e-6 = 9.5e-6 <= 1e-5 |V|.  At 8 four points of the outer-rows case broke the bar
 * (1.35e-5 |V|).  3.2e-4 of cfg3's points take the second pass (3e-5 at 8). */
#define PAIR_RULE 3.0f
__device__ __forceinline__ pk2 pair_eval(float px, float py, int i, int j, f4a16 xs, f4a16 xd, f4a16 ys, f4a16 yd, const pk2 (&q)[4][4], bool &again)
{
    const PairW w = pair_weights_lagrange(px, py, i, j, xs, xd, ys, yd);
    pk2 rw[4];
    float m = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) {      /* (a row's cells die with the row: M row by row, not from sixteen live cells at the end) */
        rw[r] = pair_row(w, q[r][0], q[r][1], q[r][2], q[r][3]);
        /* M: the central rows' largest |cell| as it is (their y weights are of order one), an OUTER row's scaled by |its y weight| (<= 0.075 on a uniform axis, whatever
         * the axis gives otherwise): what a row can put into the sum -- and into its rounding error -- is |w_y| x its largest cell (round 6: until then the outer rows were
         * not looked at, and large cells of alternating sign there could cancel unseen next to calm central rows) */
        float mr = fmaxf(fmaxf(fabsf(q[r][0].x), fabsf(q[r][0].y)), fmaxf(fabsf(q[r][1].x), fabsf(q[r][1].y)));
        mr = fmaxf(mr, fmaxf(fmaxf(fabsf(q[r][2].x), fabsf(q[r][2].y)), fmaxf(fabsf(q[r][3].x), fabsf(q[r][3].y))));
        if (r == 0) mr *= fabsf(w.y0);
        if (r == 3) mr *= fabsf(w.y3);
        m = fmaxf(m, mr);
    }
    const pk2 s = pair_cols(w, rw[0], rw[1], rw[2], rw[3]);
    again = !(fmaxf(fabsf(s.x), fabsf(s.y)) * PAIR_RULE >= m);      /* (also when something is not finite) */
    return s;
}
template <class A, bool AOS>
__device__ __forceinline__ void p_irgdint_3_w_pair(const A &Z1, const A &Z2, float px, float py, const float *ax, const float *ay,
                                                   const float *cx, const float *cy, int ni, int j1, int j2, int wrap, float &r1, float &r2)
{
    const int nnj = j2 - j1 + 1;
    int i = min(ni - 2 + wrap, max(1, max(2 - wrap, (int)px)));
    int j = min(j2 - 2, max(j1 + 1, (int)py));
    int im1, ip1, ip2; float x1, x2, x3, x4;
    irr_cols(ax, ni, wrap, i, 2, im1, ip1, ip2, x1, x2, x3, x4);
    const float *b = ay - j1;
    const double x = (double)(x2 + (x3 - x2) * (px - (float)i));
    const double y = (double)(b[j] + (b[j + 1] - b[j]) * (py - (float)j));
    const float y1 = b[j - 1], y2 = b[j], y3 = b[j + 1];
    const NewtonW wx = newton_w(coef<AOS>(cx, 0, i - 1, ni), coef<AOS>(cx, 1, i - 1, ni), coef<AOS>(cx, 2, i - 1, ni), coef<AOS>(cx, 3, i - 1, ni),
                                coef<AOS>(cx, 4, i - 1, ni), coef<AOS>(cx, 5, i - 1, ni), x - (double)x1, x - (double)x2, x - (double)x3);
    const NewtonW wy = newton_w(coef<AOS>(cy, 0, j - j1, nnj), coef<AOS>(cy, 1, j - j1, nnj), coef<AOS>(cy, 2, j - j1, nnj), coef<AOS>(cy, 3, j - j1, nnj),
                                coef<AOS>(cy, 4, j - j1, nnj), coef<AOS>(cy, 5, j - j1, nnj), y - (double)y1, y - (double)y2, y - (double)y3);
    /* (the two or three columns at the longitude seam: REAL*8 throughout, as before round 5 -- the weights as the four values' coefficients, the rows through two
     * accumulators) */
    const double wr[4] = {wy.w0, wy.w1, wy.w2, wy.g};
    double su = 0.0, sv = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double z1, z2, z3, z4;
        row_taps(Z1, im1, i, ip1, ip2, j - 1 + r, z1, z2, z3, z4);
        su = fma(wr[r], fma(wx.g, z4, fma(wx.w2, z3, fma(wx.w1, z2, wx.w0 * z1))), su);
        row_taps(Z2, im1, i, ip1, ip2, j - 1 + r, z1, z2, z3, z4);
        sv = fma(wr[r], fma(wx.g, z4, fma(wx.w2, z3, fma(wx.w1, z2, wx.w0 * z1))), sv);
    }
    r1 = (float)su; r2 = (float)sv;
}
/* The same away from the longitude seam (i - 1 .. i + 2 consecutive: all but two or three source columns), with EVERY load of the point issued before the
 * first is used: the axis entries, the two coefficient records and the eight stencil rows of the pair depend on (i, j) only.  Written in program order
 * (x, y from the axes -> coefficients -> rows) the compiler kept four dependent memory round trips per point behind the divergent seam branches;
 * k_pts2 is bound by that chain (a wave lived 5 us for 375 instructions), not by its gathers' bandwidth or its arithmetic. */
__device__ __forceinline__ void p_irgdint_3_w_pair_inner(const float *z1f, const float *z2f, float px, float py, const float *xrec8, const float *yrec8,
                                                         const float *cx8, const float *cy8, int ni, int j1, int i, int j, float &r1, float &r2)
{
    const f4a16 xs = *(const f4a16 *)(xrec8 + (size_t)(i - 1) * 8), xd = *(const f4a16 *)(xrec8 + (size_t)(i - 1) * 8 + 4);
    const f4a16 ys = *(const f4a16 *)(yrec8 + (size_t)(j - j1) * 8), yd = *(const f4a16 *)(yrec8 + (size_t)(j - j1) * 8 + 4);
    const size_t o0 = (size_t)(j - 1 - j1) * (size_t)ni + (size_t)(i - 2);
    f4u u[4], v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { u[r] = *(const f4u *)(z1f + o0 + (size_t)r * ni); v[r] = *(const f4u *)(z2f + o0 + (size_t)r * ni); }
    pk2 q[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) { q[r][0] = pk2{u[r].x, v[r].x}; q[r][1] = pk2{u[r].y, v[r].y}; q[r][2] = pk2{u[r].z, v[r].z}; q[r][3] = pk2{u[r].w, v[r].w}; }
    bool again;
    pk2 s = pair_eval(px, py, i, j, xs, xd, ys, yd, q, again);
    if (again) s = pair_eval_real8<const float *>(z1f + o0, z2f + o0, 1, ni, xrec8 + (size_t)(i - 1) * 8, yrec8 + (size_t)(j - j1) * 8, cx8 + (size_t)(i - 1) * 8, cy8 + (size_t)(j - j1) * 8, px, py, i, j);
    r1 = s.x; r2 = s.y;
}
/* ez_irgdint_3_wnnc.inc:20-246 (ay: 4-entry strip latitudes indexed from j1) */
template <class A> __device__ __forceinline__ float p_irgdint_3_wnnc(const A &Z, float px, float py, const float *ax, const float *ay4,
                                                     int ni, int j1, int j2, int wrap)
{
    int i = min(ni - 2 + wrap, max(1, max(2 - wrap, (int)px)));
    int j = min(j2 - 2, max(j1 + 1, (int)py));
    int im1, ip1, ip2; float fx1, fx2, fx3, fx4;
    irr_cols(ax, ni, wrap, i, 1, im1, ip1, ip2, fx1, fx2, fx3, fx4);
    double x1 = fx1, x2 = fx2, x3 = fx3, x4 = fx4;
    const float *b = ay4 - j1;
    double x = x2 + (x3 - x2) * (double)(px - (float)i);
    double y = (double)(b[j] + (b[j + 1] - b[j]) * (py - (float)j));
    double c1 = 1.0 / (x2 - x1), c2 = 1.0 / (x3 - x1), c3 = 1.0 / (x3 - x2);
    double c4 = 1.0 / (x4 - x1), c5 = 1.0 / (x4 - x2), c6 = 1.0 / (x4 - x3);
    double y1 = b[j - 1], y2 = b[j], y3 = b[j + 1], y4 = b[j + 2];
    double bb[4];
    for (int r = 0; r < 4; r++) {
        int jj = j - 1 + r;
        double z1, z2, z3, z4;
        row_taps(Z, im1, i, ip1, ip2, jj, z1, z2, z3, z4);
        double a2 = d_fa2(c1, z1, z2);
        double a3 = d_fa3(c1, c2, c3, z1, z2, z3);
        double a4 = d_fa4(c1, c2, c3, c4, c5, c6, z1, z2, z3, z4);
        bb[r] = d_fa(z1, a2, a3, a4, x, x1, x2, x3);
    }
    double d1 = 1.0 / (y2 - y1), d2 = 1.0 / (y3 - y1), d3 = 1.0 / (y3 - y2);
    double d4 = 1.0 / (y4 - y1), d5 = 1.0 / (y4 - y2), d6 = 1.0 / (y4 - y3);
    double b12 = d_fa2(d1, bb[0], bb[1]);
    double b13 = d_fa3(d1, d2, d3, bb[0], bb[1], bb[2]);
    double b14 = d_fa4(d1, d2, d3, d4, d5, d6, bb[0], bb[1], bb[2], bb[3]);
    return (float)d_fa(bb[0], b12, b13, b14, y, y1, y2, y3);
}

/* c_gdinterp dispatch (src/interp/gdinterp.c:133-309) for one point */
template <class A>
__device__ __noinline__ float gdinterp_point(const ezhip_pts_plan &p, const A &Z, int degree, float px, float py)
{
    if (p.irregular) {
        switch (degree) {
        case 0: return p_rgdint_0(Z, px, py, p.ni, p.j1, p.j2);
        case 1: return p.wrap == 0 ? p_irgdint_1_nw(Z, px, py, p.ax, p.ay, p.ni, p.nj)
                                   : p_irgdint_1_w(Z, px, py, p.ax, p.ay, p.ni, p.j1, p.j2, p.wrap);
        default: return p.wrap == 0 ? p_irgdint_3_nw(Z, px, py, p.ax, p.ay, p.ncx, p.ncy, p.i1, p.i2, p.j1, p.j2)
                                    : p_irgdint_3_w(Z, px, py, p.ax, p.ay, p.ncx, p.ncy, p.ni, p.j1, p.j2, p.wrap);
        }
    }
    switch (degree) {
    case 0: return p_rgdint_0(Z, px, py, p.ni, p.j1, p.j2);
    case 1: return p.wrap == 2 ? p_rgdint_1_w(Z, px, py, p.ni, p.j1, p.j2, p.wrap)
                               : p_rgdint_1_nw(Z, px, py, p.ni, p.j1, p.j2);
    default: return p.wrap == 0 ? p_rgdint_3_nw(Z, px, py, p.ni, p.j1, p.j2)
                                : p_rgdint_3_w(Z, px, py, p.ni, p.j1, p.j2, p.wrap, 0);
    }
}

/* polar strip interpolation of one point: ez_corrval_aunord.c:28-117 / ez_corrval_ausud.c:30-137 */
template <class A>
__device__ __noinline__ float strip_point(const ezhip_pts_plan &p, const A &Z, int north, float px, float py)
{
    const int j1s = north ? p.j2 - 2 : p.j1 - 1, j2s = j1s + 3;
    if (p.degree == 3) {
        if (p.irregular) return p_irgdint_3_wnnc(Z, px, py, p.ax, north ? p.ay4_n : p.ay4_s, p.ni, j1s, j2s, p.wrap);
        return p_rgdint_3_w(Z, px, py, p.ni, j1s, j2s, p.wrap, 1);
    }
    if (north) {
        /* ty = y - (j2 - 3): strip rows renumbered 1..4.  FieldAcc rows are absolute, so evaluate with
         * absolute row bounds j2-2 .. j2+1 and shift the fractional coordinate back. */
        float ty = (float)((double)py - (1.0 * (p.j2 - 3)));
        /* rows 1..4 of the strip == absolute rows j2-2..j2+1: remap through an offset accessor */
        struct Off { const A &Z; int off; __device__ float operator()(int i, int j) const { return Z(i, j + off); } } ZO{Z, p.j2 - 3};
        if (p.degree == 1) return p_rgdint_1_w(ZO, px, ty, p.ni, 1, 4, p.wrap);
        return p_rgdint_0(ZO, px, ty, p.ni, 1, 4);
    }
    if (p.degree == 1) return p_rgdint_1_w(Z, px, py, p.ni, j1s, j2s, p.wrap);
    return p_rgdint_0(Z, px, py, p.ni, j1s, j2s);
}

/* The leaf kernel of the launch as a compile-time choice (gdinterp.c:133-309 dispatch, uniform per launch):
 * one inlined leaf per k_pts instantiation instead of a call into a function that holds all nine (that version
 * needed 224 B of scratch per lane and 102 VGPRs).  The rare paths -- polar strips, re-interpolated
 * extrapolation -- stay out of line. */
enum { PK_RGD0 = 0, PK_RGD1_NW, PK_RGD1_W, PK_RGD3_NW, PK_RGD3_W, PK_IRGD1_NW, PK_IRGD1_W, PK_IRGD3_NW, PK_IRGD3_W, PK_COUNT };
template <int KIND, class A>
__device__ __forceinline__ float leaf_point(const ezhip_pts_plan &p, const A &Z, float px, float py)
{
    if (KIND == PK_RGD0) return p_rgdint_0(Z, px, py, p.ni, p.j1, p.j2);
    if (KIND == PK_RGD1_NW) return p_rgdint_1_nw(Z, px, py, p.ni, p.j1, p.j2);
    if (KIND == PK_RGD1_W) return p_rgdint_1_w(Z, px, py, p.ni, p.j1, p.j2, p.wrap);
    if (KIND == PK_RGD3_NW) return p_rgdint_3_nw(Z, px, py, p.ni, p.j1, p.j2);
    if (KIND == PK_RGD3_W) return p_rgdint_3_w(Z, px, py, p.ni, p.j1, p.j2, p.wrap, 0);
    if (KIND == PK_IRGD1_NW) return p_irgdint_1_nw(Z, px, py, p.ax, p.ay, p.ni, p.nj);
    if (KIND == PK_IRGD1_W) return p_irgdint_1_w(Z, px, py, p.ax, p.ay, p.ni, p.j1, p.j2, p.wrap);
    if (KIND == PK_IRGD3_NW) return p_irgdint_3_nw<A, true>(Z, px, py, p.ax, p.ay, p.ncx8, p.ncy8, p.i1, p.i2, p.j1, p.j2);
    return p_irgdint_3_w<A, true>(Z, px, py, p.ax, p.ay, p.ncx8, p.ncy8, p.ni, p.j1, p.j2, p.wrap);
}
static int pts_kind(const ezhip_pts_plan *p)
{
    if (p->degree == 0) return PK_RGD0;
    if (p->irregular) return p->degree == 1 ? (p->wrap == 0 ? PK_IRGD1_NW : PK_IRGD1_W) : (p->wrap == 0 ? PK_IRGD3_NW : PK_IRGD3_W);
    return p->degree == 1 ? (p->wrap == 2 ? PK_RGD1_W : PK_RGD1_NW) : (p->wrap == 0 ? PK_RGD3_NW : PK_RGD3_W);
}

/* zone of a point (ez_defzones.c:25-113 and the order ez_corrval.c:121-145 / ez_corrvec.c:24-48 apply them in) */
enum { PZ_NORMAL = 0, PZ_FILL, PZ_REINTERP, PZ_STRIP_S, PZ_STRIP_N, PZ_POLE_S, PZ_POLE_N };
__device__ __forceinline__ int pts_zone(int zones, int ni, int nj, int j1, int j2, float ypole_n, float ypole_s, int vector_mode,
                                        int degre_extrap, float px, float py)
{
    if (zones == 2) {                                     /* EZ_EXTRAP: ez_defzone_dehors.c:63-74 */
        int ix = (int)((double)px + 0.5), iy = (int)((double)py + 0.5);
        bool dehors = ix < 1 || iy < 1 || ix > ni || iy > nj;
        return !dehors ? PZ_NORMAL : (degre_extrap >= 4 ? PZ_FILL : PZ_REINTERP);
    }
    if (zones == 1) {                                     /* EZ_NO_EXTRAP */
        bool au_n = (int)py > (j2 - 2);                   /* ez_defzone_nord.c:41-49 */
        bool au_s = (int)py < (j1 + 1);                   /* ez_defzone_sud.c:42-50 */
        if (!vector_mode) {                               /* AU_NORD, AU_SUD, POLE_NORD, POLE_SUD in that order: last writer wins */
            if (fabs((double)(py - ypole_s)) < 1.0e-3) return PZ_POLE_S;
            if (fabs((double)(py - ypole_n)) < 1.0e-3) return PZ_POLE_N;
        }
        return au_s ? PZ_STRIP_S : (au_n ? PZ_STRIP_N : PZ_NORMAL);
    }
    return PZ_NORMAL;
}

/* Main kernel: ONE inlined leaf, plan members straight from the kernel-argument segment (global loads, no scratch:
 * the version that also called the out-of-line strip / re-interpolation code kept the plan in scratch and reached its
 * tables through 100 flat loads).  Points of the polar strips and re-interpolated extrapolation points are left to
 * k_pts_special. */
/* XCD-aware order of the point blocks: thread blocks are dealt round-robin to the 8 XCDs, each with its own L2; XCD k takes the k-th CONTIGUOUS eighth
 * of the target points, so that its gathers stay inside an eighth of the source (+ what the rotation spreads) instead of all of it: with the linear
 * order every L2 held the whole cfg3 source pair (measured: 370 MB fetched per 8 M point pairs for 26 MB of sources + 192 MB of x, y, matrices;
 * L2 hit rate of the gathers 64 %) */
__device__ __forceinline__ unsigned pts_block(unsigned b, unsigned nb)
{
    const unsigned full = nb & ~7u;
    return b < full ? (b & 7u) * (full >> 3) + (b >> 3) : b;
}

/* one point of the scalar per-point path.  zone_known >= 0: the caller has the zone already */
template <int KIND>
__device__ __forceinline__ void pts1_point(const ezhip_pts_plan &p, float *__restrict__ zout, const float *__restrict__ zin, float px, float py, int n,
                                           int *__restrict__ special_list, unsigned *__restrict__ special_count, int zone_known, const float *__restrict__ polevals)
{
    const size_t o = p.out_idx ? (size_t)p.out_idx[n] : (size_t)n;        /* Yin-Yang point lists write straight to their target positions */
    const PlainAcc ZP{zin, p.ni, p.j1};
    const int zone = zone_known >= 0 ? zone_known : pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, px, py);
    const bool pole_later = p.pv_out != nullptr && (zone == PZ_POLE_S || zone == PZ_POLE_N);
    if (zone == PZ_NORMAL) { if (!p.only_special) zout[o] = leaf_point<KIND>(p, ZP, px, py); }
    else if (zone == PZ_FILL) zout[o] = *p.fill;
    else if (pole_later) { }
    else if (zone == PZ_POLE_S) zout[o] = polevals[1];
    else if (zone == PZ_POLE_N) zout[o] = polevals[0];
    /* strip / re-interpolated points: appended to the launch's list, one atomic per wave that has any (special_list == nullptr: a later field of a batch, listed already) */
    const bool sp = (zone == PZ_REINTERP || zone == PZ_STRIP_S || zone == PZ_STRIP_N || pole_later) && special_list != nullptr;
    const unsigned long long m = __ballot(sp);
    if (sp) {
        const int lane = (int)__lane_id(), leader = __ffsll((long long)m) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(special_count, (unsigned)__popcll(m));
        base = (unsigned)__shfl((int)base, leader, 64);
        special_list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = n;
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_pts(ezhip_pts_plan p, float *__restrict__ zout, const float *__restrict__ zin,
                                             const float *__restrict__ xs, const float *__restrict__ ys, int npts,
                                             int *__restrict__ special_list, unsigned *__restrict__ special_count)
{
    unsigned boff = 0;
    if (p.pv_out) {            /* the field's pole values: two producer blocks at the head of this launch; whoever reads them is in the next one */
        if (blockIdx.x < 2) {
            __shared__ __attribute__((aligned(16))) float pv_lds[2048 + 4];
            const float *row = blockIdx.x == 0 ? zin + (size_t)(p.pv_nj - 1) * p.ni : zin;
            const float v = block_poleval(row, p.ni, p.pole_weighted, p.ax, pv_lds, 2048);
            if (threadIdx.x == 0) p.pv_out[blockIdx.x] = v;
            return;
        }
        boff = 2;
    }
    int n = (blockIdx.x - boff) * 256 + threadIdx.x;
    if (n >= npts) return;
    pts1_point<KIND>(p, zout, zin, xs[n], ys[n], n, special_list, special_count, -1, p.polevals);
}

/* The two components of a wind pair in one pass (c_ezuvint on the per-point path): x, y, zone test, indices and weights are
 * shared (the leaf is inlined twice and the compiler merges everything that does not depend on the field). */
/* waves_per_eu(6, 8): the irregular bicubic pair otherwise takes 118 VGPRs (4 waves per SIMD) and its gathers stop overlapping its
 * arithmetic: 80 VGPRs without spilling, -10 us per cfg3 pair.  (Lagrange weights shared by the two components -- the Newton tables'
 * reciprocals are the Lagrange denominators -- cut the VALU work from 540 to 310 instructions per wave and were measured SLOWER at equal
 * occupancy, 172 - 186 against 165 - 169 us: the kernel is bound by its gathers, not by its arithmetic; the reference's Newton form stays.) */
template <int KIND, bool LITERAL>
__device__ __forceinline__ void pts2_point(const ezhip_pts_plan &p, float *__restrict__ zout1, float *__restrict__ zout2,
                                           const float *__restrict__ zin1, const float *__restrict__ zin2,
                                           const float *__restrict__ xs, const float *__restrict__ ys, int n,
                                           int *__restrict__ special_list, unsigned *__restrict__ special_count);
template <int KIND, bool LITERAL>
__device__ __forceinline__ void pts2_body(ezhip_pts_plan p, float *__restrict__ zout1, float *__restrict__ zout2,
                                              const float *__restrict__ zin1, const float *__restrict__ zin2,
                                              const float *__restrict__ xs, const float *__restrict__ ys, int npts,
                                              int *__restrict__ special_list, unsigned *__restrict__ special_count, unsigned boff)
{
    int n;
    if (p.tile_ni > 0) {
        /* 2-D order: a wave = a patch of target points (16 x 4 since the end of round 3; 8 x 8 before).  The TCP (vector L1) looks up about one cache line per cycle; 64 consecutive points of a
         * target row touch ~10 lines per stencil-row load when the source is rotated, an 8 x 8 patch 2 - 3.  The four waves of a block sit side by side:
         * their 32-byte row pieces of x, y and the outputs make whole 128-byte lines */
        /* p.tile_shape: 1 (default) = a wave 16 x 4, a block 64 x 4; 0 = a wave 8 x 8, a block 32 x 8; 2 = a wave 4 x 16, a block 16 x 16; 3 .. 5 below */
        const unsigned bw = p.tile_shape == 1 ? 64u : p.tile_shape == 2 ? 16u : p.tile_shape == 3 ? 128u : p.tile_shape == 5 ? 16u : 32u;
        const unsigned tpr = ((unsigned)p.tile_ni + bw - 1u) / bw, b = p.xcd_order ? pts_block(blockIdx.x - boff, gridDim.x - boff) : blockIdx.x - boff;
        const unsigned by = b / tpr, bx = b - by * tpr, t = threadIdx.x;
        unsigned cx_ = bx * 32u + (t >> 6) * 8u + (t & 7u), cy_ = by * 8u + ((t >> 3) & 7u);
        if (p.tile_shape == 1) { cx_ = bx * 64u + (t >> 6) * 16u + (t & 15u); cy_ = by * 4u + ((t >> 4) & 3u); }
        else if (p.tile_shape == 2) { cx_ = bx * 16u + (t >> 6) * 4u + (t & 3u); cy_ = by * 16u + ((t >> 2) & 15u); }
        else if (p.tile_shape == 3) { cx_ = bx * 128u + (t >> 6) * 32u + (t & 31u); cy_ = by * 2u + ((t >> 5) & 1u); }                       /* a wave 32 x 2, a block 128 x 2 */
        else if (p.tile_shape == 4) { cx_ = bx * 32u + ((t >> 6) & 1u) * 16u + (t & 15u); cy_ = by * 8u + (t >> 7) * 4u + ((t >> 4) & 3u); }    /* a wave 16 x 4, a block 32 x 8 (2 x 2 waves) */
        else if (p.tile_shape == 5) { cx_ = bx * 16u + (t & 15u); cy_ = by * 16u + (t >> 6) * 4u + ((t >> 4) & 3u); }                            /* a wave 16 x 4, a block 16 x 16 (waves stacked) */
        if (cx_ >= (unsigned)p.tile_ni || cy_ >= (unsigned)p.tile_nj) return;
        n = (int)(cy_ * (unsigned)p.tile_ni + cx_);
    } else n = (int)(p.xcd_order ? pts_block(blockIdx.x - boff, gridDim.x - boff) : blockIdx.x - boff) * 256 + threadIdx.x;
    if (n >= npts) return;
    pts2_point<KIND, LITERAL>(p, zout1, zout2, zin1, zin2, xs, ys, n, special_list, special_count);
}
/* one point pair of k_pts2 / k_pts2_irgd3w (and of the tiles k_uvt hands back) */
template <int KIND, bool LITERAL>
__device__ __forceinline__ void pts2_point(const ezhip_pts_plan &p, float *__restrict__ zout1, float *__restrict__ zout2,
                                           const float *__restrict__ zin1, const float *__restrict__ zin2,
                                           const float *__restrict__ xs, const float *__restrict__ ys, int n,
                                           int *__restrict__ special_list, unsigned *__restrict__ special_count)
{
    const float px = xs[n], py = ys[n];
    const size_t o = p.out_idx ? (size_t)p.out_idx[n] : (size_t)n;
    const PlainAcc Z1{zin1, p.ni, p.j1}, Z2{zin2, p.ni, p.j1};
    const int zone = pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, px, py);
    /* the point's wind matrix is fetched first: nothing depends on it until the store, its latency hides behind the interpolation
     * (fetched where it is applied it cost +37 us per cfg3 pair, as much as the separate k_wind_apply pass) */
    wm_f2 wlo = {1.0f, 0.0f}, whi = {0.0f, 1.0f};
    /* (the empty asm pins the load here -- and makes the compiler wait for it here: one memory round trip before the interpolation starts.  The
     * all-loads-first form below issues it with the point's other loads instead) */
    constexpr bool FRONT = KIND == PK_IRGD3_W && !LITERAL;
    if (p.wind_M && !FRONT) { wind_m_load(p.wind_M, p.wind_M_half, o, wlo, whi); asm volatile("" : "+v"(wlo), "+v"(whi)); }
    if (zone == PZ_NORMAL || zone == PZ_FILL) {
        float a, b;
        if (zone == PZ_FILL) a = b = *p.fill;
        else if (KIND == PK_IRGD3_W && !LITERAL) {
            const int i = min(p.ni - 2 + p.wrap, max(1, max(2 - p.wrap, (int)px))), j = min(p.j2 - 2, max(p.j1 + 1, (int)py));
            const bool seam = i <= 1 || i >= p.ni - 1;                         /* (wrap 0: i stays in 2 .. ni - 2) */
            /* in flight with the loads of the interpolation; unconditional (a conditional load is merged with the identity at the join: a use, hence a
             * wait, right behind it): without a matrix a readable dummy address */
            wind_m_load(p.wind_M ? p.wind_M : (const void *)p.ncx8, p.wind_M_half, p.wind_M ? o : (size_t)0, wlo, whi);
            /* (per LANE: the two forms round differently -- the seam's weights come from the Newton tables -- and which waves a point shares differs between
             * the kernels of the pair path; a point's form must not depend on its neighbours) */
            if (!seam) p_irgdint_3_w_pair_inner(zin1, zin2, px, py, p.xrec8, p.yrec8, p.ncx8, p.ncy8, p.ni, p.j1, i, j, a, b);
            else p_irgdint_3_w_pair<PlainAcc, true>(Z1, Z2, px, py, p.ax, p.ay, p.ncx8, p.ncy8, p.ni, p.j1, p.j2, p.wrap, a, b);
        }
        else { a = leaf_point<KIND>(p, Z1, px, py); b = leaf_point<KIND>(p, Z2, px, py); }
        if (p.wind_M) {                       /* the wind chain of the grid pair (k_wind_apply), here instead of a pass of its own */
            const float u = a, v = b;
            wind_m_apply(wlo, whi, p.wind_M_half, u, v, p.wind_dst_rot, a, b);
        }
        zout1[o] = a; zout2[o] = b;
    }
    const bool sp = (zone == PZ_REINTERP || zone == PZ_STRIP_S || zone == PZ_STRIP_N) && special_list != nullptr;      /* (nullptr: the host already knows this set's special points) */
    const unsigned long long m = __ballot(sp);
    if (sp) {
        const int lane = (int)__lane_id(), leader = __ffsll((long long)m) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(special_count, (unsigned)__popcll(m));
        base = (unsigned)__shfl((int)base, leader, 64);
        special_list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = n;
    }
}

template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_pts2(ezhip_pts_plan p, float *__restrict__ zout1, float *__restrict__ zout2,
                                              const float *__restrict__ zin1, const float *__restrict__ zin2,
                                              const float *__restrict__ xs, const float *__restrict__ ys, int npts,
                                              int *__restrict__ special_list, unsigned *__restrict__ special_count)
{
    pts2_body<KIND, true>(p, zout1, zout2, zin1, zin2, xs, ys, npts, special_list, special_count, 0u);
}
/* winds from an irregular source with wrap (cfg3): the regrouped Newton form with every load of a point in flight at once: more registers per
 * lane (no occupancy floor), half the dependent memory round trips */
__global__ __launch_bounds__(256) void k_pts2_irgd3w(ezhip_pts_plan p, float *__restrict__ zout1, float *__restrict__ zout2,
                                              const float *__restrict__ zin1, const float *__restrict__ zin2,
                                              const float *__restrict__ xs, const float *__restrict__ ys, int npts,
                                              int *__restrict__ special_list, unsigned *__restrict__ special_count)
{
    /* the pair's synthetic polar wind rows: two producer blocks at the head of this launch (they were a 2-block kernel on a side stream, forked from and
     * joined into this stream with events: ~7 us of launch gaps per wind pair around a 107 us kernel).  Only the special points read the rows, and those
     * are the next kernel's */
    unsigned boff = 0;
    if (p.pw_out) {
        if (blockIdx.x < 2) {
            __shared__ __attribute__((aligned(16))) float pw_lds[2048 + 4];
            polar_wind_body<2048>(blockIdx.x == 0, p.pw_out, zin1, zin2, p.pw_plon2, p.ni, p.nj, p.pw_xg4_n, p.pw_xg4_s, p.pw_weighted, p.pw_ax, pw_lds); return;
        }
        boff = 2;
    }
    pts2_body<PK_IRGD3_W, false>(p, zout1, zout2, zin1, zin2, xs, ys, npts, special_list, special_count, boff);
}

/* ---- k_uvt: the wind pair from an irregular (rotated) source with its stencil tiles STAGED IN LDS -------------------------------------------
 * k_pts2_irgd3w gathers 17 times per point pair through the vector L1 (8 stencil rows, 6 coefficient pieces, x, y, the wind matrix): 404 cache
 * accesses of 64 bytes per wave, the L1's 64 B / clk is the kernel's bound (profiles/r03_experiments.txt).  Here a thread block takes a 32 x UVT_TH
 * tile of the target; the source window its normal points' stencils touch (known per tile from the set's located x, y: k_uvt_bbox, once per grid set)
 * comes in with coalesced loads -- both components side by side as float2 cells -- together with the columns' and rows' axis / Newton-coefficient records
 * (48 bytes each: ax(i-1 .. i+2), c1 .. c6); every point then reads its 16 cells and its two records from LDS.  Through the vector L1 go only the
 * streams (x, y, matrix in; u, v out: whole 128-byte row pieces) and the staging.  Tiles that do not qualify (the longitude seam; windows beyond UVT_CAP
 * cells next to the rotated poles) take the gathering path point by point; polar-strip and re-interpolated points stay with k_pts_special2c.
 * The arithmetic is p_irgdint_3_w_pair_inner's, operation for operation: results are bit-identical to k_pts2_irgd3w's. */
#define UVT_CAP_DEFAULT 2560                           /* staged cells per tile */
#define UVT_REC_MAX 128                                /* records (x + y) per tile, 80 bytes each */
#define UVT_ALL_NORMAL 0x40000000                      /* tile table, .w: every point of the tile is a main-zone point */
#define UVT_H_MASK 0xFFFF
__device__ __forceinline__ int uvt_wave_min(int v) { for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ int uvt_wave_max(int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64)); return v; }
/* tile (TW x TH target points, a thread block of 256: thread t takes column t % TW, rows t / TW + k * 256 / TW) */
template <int TW, int TH> struct uvt_geom {
    static constexpr int PPT = TW * TH / 256, RSTEP = 256 / TW;
    static_assert(TW * TH % 256 == 0 && (TW == 32 || TW == 64), "k_uvt tile shape");
};
/* per tile {i0, j0, W, H}: source columns i0 .. i0 + W - 1 (1-based) and rows j0 .. j0 + H - 1 under the stencils of the tile's NORMAL points;
 * W = 0: hand the tile to the gathering path; W < 0: nothing in the tile is this kernel's (only special points) */
template <int TW, int TH>
__global__ __launch_bounds__(256) void k_uvt_bbox(ezhip_pts_plan p, const float *__restrict__ xs, const float *__restrict__ ys, int4 *__restrict__ tiles, int cap, int recmax)
{
    typedef uvt_geom<TW, TH> G;
    __shared__ int red[4][7];
    const unsigned tpr = ((unsigned)p.tile_ni + TW - 1u) / TW, b = blockIdx.x, by = b / tpr, bx = b - by * tpr, t = threadIdx.x;
    const unsigned cx = bx * TW + (t % TW);
    int imin = 0x7fffffff, imax = -1, jmin = 0x7fffffff, jmax = -1, seam = 0, mine = 0, other = 0;
#pragma unroll
    for (int k = 0; k < G::PPT; k++) {
        const unsigned cy = by * TH + t / TW + (unsigned)(G::RSTEP * k);
        if (cx >= (unsigned)p.tile_ni || cy >= (unsigned)p.tile_nj) continue;
        const size_t n = (size_t)cy * p.tile_ni + cx;
        const float px = xs[n], py = ys[n];
        const int zone = pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, px, py);
        if (zone == PZ_FILL) mine = 1;
        if (zone != PZ_NORMAL) other = 1;
        if (zone != PZ_NORMAL) continue;
        mine = 1;
        const int i = min(p.ni - 2 + p.wrap, max(1, max(2 - p.wrap, (int)px))), j = min(p.j2 - 2, max(p.j1 + 1, (int)py));
        seam |= (i <= 1 || i >= p.ni - 1) ? 1 : 0;
        imin = min(imin, i); imax = max(imax, i); jmin = min(jmin, j); jmax = max(jmax, j);
    }
    imin = uvt_wave_min(imin); jmin = uvt_wave_min(jmin); imax = uvt_wave_max(imax); jmax = uvt_wave_max(jmax); seam = uvt_wave_max(seam); mine = uvt_wave_max(mine); other = uvt_wave_max(other);
    if ((t & 63u) == 0) { int *r = red[t >> 6]; r[0] = imin; r[1] = imax; r[2] = jmin; r[3] = jmax; r[4] = seam; r[5] = mine; r[6] = other; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 4; w++) { imin = min(imin, red[w][0]); imax = max(imax, red[w][1]); jmin = min(jmin, red[w][2]); jmax = max(jmax, red[w][3]); seam |= red[w][4]; mine |= red[w][5]; other |= red[w][6]; }
        int4 o;
        if (!mine) o = make_int4(0, 0, -1, 0);
        else if (imax < 0) o = make_int4(0, 0, 0, 0);                                     /* fill points only: the gathering path writes them */
        else {
            const int W = imax - imin + 4, H = jmax - jmin + 4;
            const bool ok = !seam && W * H <= cap && (W - 3) + (H - 3) <= recmax;
            o = ok ? make_int4(imin - 1, jmin - 1, W, H | (other ? 0 : UVT_ALL_NORMAL)) : make_int4(0, 0, 0, 0);
        }
        tiles[b] = o;
    }
}
/* The streams of a tile -- x, y and the wind matrix's (a, b) of its points -- are the grid SET's own data, so their layout is ours: kept a second time in tile
 * order, one 12-byte record {x, y, packed rotation} per point (16 bytes with (a, b) as two REALs until round 5), [tile][k][thread].  A block then reads 12 contiguous KB (4 loads of 12 bytes per thread) where the row-major
 * arrays give it 128-byte pieces 16 KB apart in three arrays (12 loads per thread): those pieces, not the bytes, bounded the streaming side of k_uvt
 * (2.5 - 3.5 TB/s with everything else knocked out: profiles/r04_experiments.txt).  12 bytes per target point of extra HBM per grid set. */
struct __attribute__((aligned(4))) uvt_rec { float x, y; unsigned q; };      /* 12 bytes per point: x, y, the packed rotation (rot_pack; the identity without a matrix) */
template <int TW, int TH>
__global__ __launch_bounds__(256) void k_uvt_pack(ezhip_pts_plan p, const float *__restrict__ xs, const float *__restrict__ ys, uvt_rec *__restrict__ streams)
{
    typedef uvt_geom<TW, TH> G;
    const unsigned tpr = ((unsigned)p.tile_ni + TW - 1u) / TW, b = blockIdx.x, by = b / tpr, bx = b - by * tpr, t = threadIdx.x;
    const unsigned cx = bx * TW + (t % TW);
#pragma unroll
    for (int k = 0; k < G::PPT; k++) {
        const unsigned cy = by * TH + t / TW + (unsigned)(G::RSTEP * k);
        uvt_rec o{0.f, 0.f, rot_pack(1.0f, 0.0f)};
        if (cx < (unsigned)p.tile_ni && cy < (unsigned)p.tile_nj) {
            const size_t n = (size_t)cy * p.tile_ni + cx;
            o.x = xs[n]; o.y = ys[n];
            if (p.wind_M) o.q = ((const unsigned *)p.wind_M)[n];
        }
        streams[((size_t)b * G::PPT + k) * 256 + t] = o;
    }
}
/* (The set's special points -- polar strips, re-interpolated extrapolation: a fraction of a percent of the points, a chain of dependent gathers, 12 us of
 * latency per cfg3 pair as a kernel of their own behind this one -- were tried (a) on a side stream beside this kernel: fork / join events, 98.8 against
 * 87.7 us per pair; (b) as blocks at the head of this launch waiting for the polar-wind producers: the out-of-line strip / re-interpolation code they call
 * takes 146 VGPRs and a call stack, capped at 128 it spills into scratch and the WHOLE launch slows to 227 us.  They stay a kernel of their own.) */
#ifndef UVT_WAVES
#define UVT_WAVES 5
#endif
#ifndef UVT_HB_UNROLL
#define UVT_HB_UNROLL 4                                 /* points of a handed-back tile (the gathering path) in flight per thread */
#endif
#ifndef UVT_WAVES_B
#define UVT_WAVES_B 5                                   /* the batch form (c_ezuvint_batch_dev): the pair loop needs 96 registers (at 80 it spills 160 bytes: 69 - 90 us per pair
                                                         * against 50 - 60; 4 waves: the same as 5; the next pair's window prefetched into 20 registers at 4 waves: 56 - 79) */
#endif
#ifndef UVT_WAVES_W
#define UVT_WAVES_W 6                                   /* the REAL form of the wrap-around variant needs fewer registers, its records a third of the LDS */
#endif

template <bool STRIP3 = false>
__device__ __forceinline__ void special2c_body(const ezhip_pts_plan &p, float *__restrict__ zout1, float *__restrict__ zout2, const float *__restrict__ zin1, const float *__restrict__ zin2,
                                               const float *__restrict__ prow_n2, const float *__restrict__ prow_s2, unsigned blk, unsigned nblk, size_t roff = 0, int part = 0);
template <int TW, int TH, bool NW = false, bool BATCH = false>      /* BATCH: p.npairs wind pairs one after the other per tile (c_ezuvint_batch_dev): the points' x, y and rotation, the
                                                 * tile's table entry and its axis records once for all of them; NW: a source without wrap (a regional 'Z' grid): both components in the LITERAL form of ez_irgdint_3_nw.inc (REAL statement
                                                 * functions), as k_pts2<PK_IRGD3_NW> evaluates them on the set's first call */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BATCH ? UVT_WAVES_B : NW ? UVT_WAVES : UVT_WAVES_W, 8))) void k_uvt(ezhip_pts_plan p, float *__restrict__ zout1, float *__restrict__ zout2,
                                             const float *__restrict__ zin1, const float *__restrict__ zin2,
                                             const float *__restrict__ xs, const float *__restrict__ ys, const int4 *__restrict__ tiles)
{
    typedef uvt_geom<TW, TH> G;
    constexpr int PPT = G::PPT;
    extern __shared__ __attribute__((aligned(16))) float uvt_lds[];
    const int dbg = EZH_DBG(p.uvt_debug);      /* development knock-outs (EZHIP_UVT_DEBUG, develop build only): 1 skip handed-back tiles, 2 no staging loads, 4 no arithmetic, 16 no matrix */
    unsigned boff = 0;
    const int npairs = BATCH ? p.npairs : 1;
    if (p.pw_out) {           /* the pair's synthetic polar wind rows: two producer blocks at the head of the launch, as in k_pts2_irgd3w (a batch: two per pair) */
        if (blockIdx.x < 2u * (unsigned)npairs) {
            const size_t f = BATCH ? blockIdx.x >> 1 : 0;
            polar_wind_body<2048>((blockIdx.x & 1u) == 0, p.pw_out + f * (size_t)p.pair_rows_stride, zin1 + f * p.pair_in_stride, zin2 + f * p.pair_in_stride, p.pw_plon2, p.ni, p.nj, p.pw_xg4_n, p.pw_xg4_s, p.pw_weighted, p.pw_ax, uvt_lds);
            if (!NW && p.cspec_inline) {
                /* the set's special points (a few polar-strip and re-interpolated points: a chain of dependent gathers, 5 us as a launch of their own behind this kernel)
                 * right here: the northern producer has the northern rows it needs, the southern one the southern; nobody else writes these points */
                __threadfence_block();
                __syncthreads();
                special2c_body<true>(p, zout1 + f * p.pair_out_stride, zout2 + f * p.pair_out_stride, zin1 + f * p.pair_in_stride, zin2 + f * p.pair_in_stride,
                                     p.pw_out + 2 * (size_t)p.ni, p.pw_out + 3 * (size_t)p.ni, 0u, 1u, f * (size_t)p.pair_rows_stride, (blockIdx.x & 1u) ? 2 : 1);
            }
            return;      /* (the launch's dynamic LDS holds 2052 floats and more) */
        }
        boff = 2u * (unsigned)npairs;
    }
    /* (xcd_order: XCD k takes the k-th contiguous eighth of the tiles -- neighbouring tiles' windows overlap and share cache lines: one L2 then fetches them once) */
    /* (round 6) the tiles handed back to the gathering path run ~7 times as long as a staged one: the launch's first uvt_nhb blocks take them (listed behind the table), the
     * blocks in table order skip them */
    const unsigned nhb = (unsigned)p.uvt_nhb, bt = blockIdx.x - boff;
    const bool hb_block = bt < nhb;
    unsigned b = hb_block ? 0u : (p.xcd_order ? pts_block(bt - nhb, gridDim.x - boff - nhb) : bt - nhb);
    if (hb_block) b = ((const unsigned *)(tiles + (gridDim.x - boff - nhb)))[bt];
    const unsigned tpr = ((unsigned)p.tile_ni + TW - 1u) / TW, by = b / tpr, bx = b - by * tpr, t = threadIdx.x;
    const unsigned cx = bx * TW + (t % TW), cy0 = by * TH + t / TW;
    const bool okx = cx < (unsigned)p.tile_ni;
    /* the points' own streams first -- before the tile's table entry is even looked at (their addresses depend on the block index only): in flight
     * while the entry arrives and the window is staged.  (Point index 0 for the lanes beyond the target's edge: a readable address.) */
    float px[PPT], py[PPT]; wm_f2 wlo[PPT];
    const unsigned n0 = okx && cy0 < (unsigned)p.tile_nj ? cy0 * (unsigned)p.tile_ni + cx : 0u, nstep = (unsigned)G::RSTEP * (unsigned)p.tile_ni;
    if (p.uvt_streams) {              /* the set's tile-ordered copy: {x, y, rotation} of a point in one 12-byte load, a block's share contiguous */
        const uvt_rec *S = (const uvt_rec *)p.uvt_streams + (size_t)b * (PPT * 256) + t;
        /* (the rotation word rides in wlo[k].x and is unpacked where it is applied, behind the barrier: unpacked here the loads would be waited for before the window is asked for) */
#pragma unroll
        for (int k = 0; k < PPT; k++) { const uvt_rec *r = S + k * 256; px[k] = __builtin_nontemporal_load(&r->x); py[k] = __builtin_nontemporal_load(&r->y); wlo[k] = wm_f2{__uint_as_float(__builtin_nontemporal_load(&r->q)), 0.0f}; }
    } else {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const bool ok = okx && cy0 + (unsigned)(G::RSTEP * k) < (unsigned)p.tile_nj;
            const unsigned n = ok ? n0 + (unsigned)k * nstep : 0u;
            px[k] = xs[n]; py[k] = ys[n];
            wlo[k] = wm_f2{1.0f, 0.0f};
            if (p.wind_M && !(dbg & 16)) wlo[k] = p.wind_M_half ? rot_unpack(__builtin_nontemporal_load((const unsigned *)p.wind_M + n)) : __builtin_nontemporal_load((const wm_f2 *)((const char *)p.wind_M + ((size_t)n << 4)));
        }
    }
    const int4 tb = tiles[b];
    if (tb.z < 0) return;
    if (tb.z == 0) {
        if ((dbg & 1) || (nhb != 0u && !hb_block)) return;
#pragma unroll 1
        for (int f = 0; f < npairs; f++) {
            const size_t oi = BATCH ? (size_t)f * p.pair_in_stride : 0, oo = BATCH ? (size_t)f * p.pair_out_stride : 0;
#pragma unroll UVT_HB_UNROLL
            for (int k = 0; k < PPT; k++) {
                const unsigned cy = cy0 + (unsigned)(G::RSTEP * k);
                if (okx && cy < (unsigned)p.tile_nj) pts2_point<NW ? PK_IRGD3_NW : PK_IRGD3_W, NW>(p, zout1 + oo, zout2 + oo, zin1 + oi, zin2 + oi, xs, ys, (int)(cy * (unsigned)p.tile_ni + cx), nullptr, nullptr);
            }
        }
        return;
    }
    const int i0 = tb.x, j0 = tb.y, W = tb.z, H = tb.w & UVT_H_MASK, ncell = W * H;
    const bool all_normal = (tb.w & UVT_ALL_NORMAL) != 0;          /* every point of the tile lies in the main zone (k_uvt_bbox): no zone test per point */
    typedef float c2 __attribute__((ext_vector_type(2)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    c2 *cells = (c2 *)uvt_lds;
    /* NW: 5 x 16 bytes per record (REAL*8, the literal form's operands); else 2 x 16 bytes (REAL: axis entries, Lagrange denominators) */
    d2 *xr = (d2 *)(uvt_lds + 2 * ((ncell + 1) & ~1));
    const int nxr = (W - 3) * (NW ? 5 : 2), nyr = (H - 3) * (NW ? 5 : 2);
    d2 *yr = xr + nxr;
    float *const zout1_0 = zout1, *const zout2_0 = zout2;
    const float *const zin1_0 = zin1, *const zin2_0 = zin2;
#pragma unroll 1
    for (int f = 0; f < npairs; f++) {                          /* (one pass unless BATCH) */
    if (BATCH) {
        zout1 = zout1_0 + (size_t)f * p.pair_out_stride; zout2 = zout2_0 + (size_t)f * p.pair_out_stride;
        zin1 = zin1_0 + (size_t)f * p.pair_in_stride; zin2 = zin2_0 + (size_t)f * p.pair_in_stride;
        if (f) __syncthreads();                                 /* the pair before has been evaluated: its window may go */
        /* what depends on x, y alone (weights, cell addresses) is formed again for every pair: hoisted out of this loop for the thread's four points it does not
         * fit the registers (308 bytes of spills at 96 VGPRs); the kernel is not bound by its arithmetic */
#pragma unroll
        for (int k = 0; k < PPT; k++) asm volatile("" : "+v"(px[k]), "+v"(py[k]));
    }
    {
        const unsigned magic = 0xFFFFFFFFu / (unsigned)W + 1u;
        const float *s1 = zin1 + (size_t)(j0 - p.j1) * (size_t)p.ni + (size_t)(i0 - 1), *s2 = zin2 + (size_t)(j0 - p.j1) * (size_t)p.ni + (size_t)(i0 - 1);
        if (!(dbg & 2)) {
#pragma unroll 4
            for (int idx = (int)t; idx < ncell; idx += 256) {
                const unsigned r = __umulhi((unsigned)idx, magic), c = (unsigned)idx - r * (unsigned)W;
                const size_t off = (size_t)r * (size_t)p.ni + c;
                cells[idx] = c2{s1[off], s2[off]};
            }
        }
        if (!BATCH || f == 0) {
            const d2 *gx = NW ? (const d2 *)p.xrec10 + (size_t)i0 * 5 : (const d2 *)p.xrec8 + (size_t)i0 * 2;
            const d2 *gy = NW ? (const d2 *)p.yrec10 + (size_t)(j0 + 1 - p.j1) * 5 : (const d2 *)p.yrec8 + (size_t)(j0 + 1 - p.j1) * 2;
            for (int idx = (int)t; idx < nxr; idx += 256) xr[idx] = gx[idx];
            for (int idx = (int)t; idx < nyr; idx += 256) yr[idx] = gy[idx];
        }
    }
    __syncthreads();
    unsigned redo = 0;
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        if (!(okx && cy0 + (unsigned)(G::RSTEP * k) < (unsigned)p.tile_nj)) continue;
        const size_t n = (size_t)n0 + (size_t)k * nstep;
        const int zone = all_normal ? (int)PZ_NORMAL : pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, px[k], py[k]);
        float a, bb;
        if (zone == PZ_FILL) a = bb = *p.fill;
        else if (zone == PZ_NORMAL) {
            const int i = min(p.ni - 2 + p.wrap, max(1, max(2 - p.wrap, (int)px[k]))), j = min(p.j2 - 2, max(p.j1 + 1, (int)py[k]));
            if (dbg & 4) { a = px[k] + (float)i; bb = py[k] + (float)j; }
            else {
                const c2 *cp = cells + (j - 1 - j0) * W + (i - 1 - i0);
                c2 q[4][4];
                if (!NW && !p.uvt_read2) {
                    /* sixteen ds_read_b64 (2 LDS-array cycles each) instead of the eight ds_read2_b64 the compiler pairs them into (8 cycles each: half the bytes per
                     * cycle, MI355X_MICROARCH.md LDS table) -- the LDS array is this kernel's busiest unit (47 %).  Issued by hand: the compiler does not see them, the
                     * wait below names all sixteen results */
                    const unsigned a0 = lds_addr_of(cp), a1 = a0 + 8u * (unsigned)W, a2 = a1 + 8u * (unsigned)W, a3 = a2 + 8u * (unsigned)W;
#define UVT_RD4(r, A) asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24" \
                                   : "=&v"(q[r][0]), "=&v"(q[r][1]), "=&v"(q[r][2]), "=&v"(q[r][3]) : "v"(A))
                    UVT_RD4(0, a0); UVT_RD4(1, a1); UVT_RD4(2, a2); UVT_RD4(3, a3);
#undef UVT_RD4
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]), "+v"(q[1][3]),
                                                          "+v"(q[2][0]), "+v"(q[2][1]), "+v"(q[2][2]), "+v"(q[2][3]), "+v"(q[3][0]), "+v"(q[3][1]), "+v"(q[3][2]), "+v"(q[3][3]));
                } else {
#pragma unroll
                for (int r = 0; r < 4; r++) { q[r][0] = cp[r * W]; q[r][1] = cp[r * W + 1]; q[r][2] = cp[r * W + 2]; q[r][3] = cp[r * W + 3]; }
                }
                if (NW) {
                /* records {x1, x2 | x3, c1 | c2, c3 | c4, c5 | c6, c5 + c2} in REAL*8: the table's REAL entries converted once per grid, not per point */
                const d2 *xq = xr + (i - 1 - i0) * 5, *yq = yr + (j - 1 - j0) * 5;
                const d2 xa = xq[0], xb = xq[1], xc = xq[2], xd = xq[3], xe = xq[4];
                const d2 ya = yq[0], yb = yq[1], yc = yq[2], yd = yq[3], ye = yq[4];
                const float fx2 = (float)xa.y, fx3 = (float)xb.x, fy2 = (float)ya.y, fy3 = (float)yb.x;      /* (exact: they were REAL) */
                const double x = (double)(fx2 + (fx3 - fx2) * (px[k] - (float)i));
                const double y = (double)(fy2 + (fy3 - fy2) * (py[k] - (float)j));
#define UV_RF(e) ((double)(float)(e))
                    float res[2];
#pragma unroll
                    for (int comp = 0; comp < 2; comp++) {
                        double br[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const double z1 = (double)(comp ? q[r][0].y : q[r][0].x), z2 = (double)(comp ? q[r][1].y : q[r][1].x), z3 = (double)(comp ? q[r][2].y : q[r][2].x), z4 = (double)(comp ? q[r][3].y : q[r][3].x);
                            const double a2 = UV_RF(d_fa2(xb.y, z1, z2));
                            const double a3 = UV_RF(d_fa3(xb.y, xc.x, xc.y, z1, z2, z3));
                            const double a4 = UV_RF(d_fa4(xb.y, xc.x, xc.y, xd.x, xd.y, xe.x, z1, z2, z3, z4));
                            br[r] = UV_RF(d_fa(z1, a2, a3, a4, x, xa.x, xa.y, xb.x));
                        }
                        const double b12 = UV_RF(d_fa2(yb.y, br[0], br[1]));
                        const double b13 = UV_RF(d_fa3(yb.y, yc.x, yc.y, br[0], br[1], br[2]));
                        const double b14 = UV_RF(d_fa4(yb.y, yc.x, yc.y, yd.x, yd.y, ye.x, br[0], br[1], br[2], br[3]));
                        res[comp] = (float)d_fa(br[0], b12, b13, b14, y, ya.x, ya.y, yb.x);
                    }
#undef UV_RF
                    a = res[0]; bb = res[1];
                } else {
                    /* p_irgdint_3_w_pair_inner's arithmetic, operation for operation, on the staged window and records */
                    const f4a16 *xq = (const f4a16 *)xr + (i - 1 - i0) * 2, *yq = (const f4a16 *)yr + (j - 1 - j0) * 2;
                    bool again;
                    const pk2 s = pair_eval(px[k], py[k], i, j, xq[0], xq[1], yq[0], yq[1], q, again);
                    if (again) redo |= 1u << k;                 /* (nearly calm under a strong stencil: once more in REAL*8, below) */
                    a = s.x; bb = s.y;
                }
            }
        } else continue;
        if (p.wind_M) {
            wm_f2 whi = wm_f2{0.0f, 1.0f};
            if (!p.wind_M_half) whi = __builtin_nontemporal_load((const wm_f2 *)((const char *)p.wind_M + (n << 4)) + 1);      /* (sets whose chain is not a pure rotation: rare) */
            const float u = a, v = bb;
            wind_m_apply(p.uvt_streams ? rot_unpack(__float_as_uint(wlo[k].x)) : wlo[k], whi, p.wind_M_half, u, v, p.wind_dst_rot, a, bb);
        }
        if (!(dbg & 32) || a == 12345.678f) { zout1[n] = a; zout2[n] = bb; }      /* (32: development, no stores) */
    }
    /* second pass: the thread's flagged points once more in REAL*8 (a few waves in a thousand have any), stored over the REAL result */
    if (!NW && __ballot(redo != 0u) != 0ull) {
        typedef __attribute__((address_space(3))) const float *ldsf;
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            if (!((redo >> k) & 1u)) continue;
            const size_t n = (size_t)n0 + (size_t)k * nstep;
            const float fx = px[k], fy = py[k];
            const int i = min(p.ni - 2 + p.wrap, max(1, max(2 - p.wrap, (int)fx))), j = min(p.j2 - 2, max(p.j1 + 1, (int)fy));
            const ldsf cu = (ldsf)(uvt_lds + 2 * ((j - 1 - j0) * W + (i - 1 - i0)));
            const ldsf xq = (ldsf)((const float *)xr + (i - 1 - i0) * 8), yq = (ldsf)((const float *)yr + (j - 1 - j0) * 8);
            const pk2 s = pair_eval_real8<ldsf>(cu, cu + 1, 2, 2 * W, xq, yq, p.ncx8 + (size_t)(i - 1) * 8, p.ncy8 + (size_t)(j - p.j1) * 8, fx, fy, i, j);
            float a = s.x, bb = s.y;
            if (p.wind_M) {
                wm_f2 whi = wm_f2{0.0f, 1.0f};
                if (!p.wind_M_half) whi = __builtin_nontemporal_load((const wm_f2 *)((const char *)p.wind_M + (n << 4)) + 1);
                const float u = a, v = bb;
                wind_m_apply(p.uvt_streams ? rot_unpack(__float_as_uint(wlo[k].x)) : wlo[k], whi, p.wind_M_half, u, v, p.wind_dst_rot, a, bb);
            }
            zout1[n] = a; zout2[n] = bb;
        }
    }
    }                                                           /* (pairs) */
}

/* ---- k_st: the SCALAR twin of k_uvt -- c_ezsint from an irregular (rotated) source, bicubic, with its stencil windows staged in LDS ------------------------
 * Same tile table (built under the scalar zone rules: the pole points are zones of their own there), the source window as float cells, the same REAL*8 axis /
 * coefficient records; every point then evaluates ez_irgdint_3_w.inc:20-235 in its LITERAL form (the statement functions fa2, fa3, fa4, fa as the reference writes
 * them: p_irgdint_3_w above, operand for operand) from LDS -- bit-identical to k_pts<PK_IRGD3_W>, which gathers the same 28 values per point through the vector L1
 * (72 us per 4000 x 2000 field from a 2560 x 1280 source).  x, y come from a tile-ordered float2 copy kept with the set (8 bytes per target point).  Handed-back
 * tiles take the gathering path point by point; pole points, polar strips and re-interpolated points are listed for k_pts_special as k_pts lists them; the field's
 * two pole values are summed by two producer blocks at the head of the launch. */
template <int TW, int TH>
__global__ __launch_bounds__(256) void k_st_pack(ezhip_pts_plan p, const float *__restrict__ xs, const float *__restrict__ ys, float2 *__restrict__ streams)
{
    typedef uvt_geom<TW, TH> G;
    const unsigned tpr = ((unsigned)p.tile_ni + TW - 1u) / TW, b = blockIdx.x, by = b / tpr, bx = b - by * tpr, t = threadIdx.x;
    const unsigned cx = bx * TW + (t % TW);
#pragma unroll
    for (int k = 0; k < G::PPT; k++) {
        const unsigned cy = by * TH + t / TW + (unsigned)(G::RSTEP * k);
        float2 o = make_float2(0.f, 0.f);
        if (cx < (unsigned)p.tile_ni && cy < (unsigned)p.tile_nj) { const size_t n = (size_t)cy * p.tile_ni + cx; o.x = xs[n]; o.y = ys[n]; }
        streams[((size_t)b * G::PPT + k) * 256 + t] = o;
    }
}
#ifndef ST_HB_UNROLL
#define ST_HB_UNROLL 1                                  /* points of a handed-back tile in flight per thread (k_st) */
#endif
#ifndef ST_WAVES
#define ST_WAVES 7
#endif
#ifndef STB_WAVES
#define STB_WAVES 4
#endif
template <int TW, int TH, bool NW, bool BATCH>      /* BATCH: nfields > 1 with the next field's window prefetched (its own instantiation: ten more registers); NW: a source without wrap (a regional 'Z' grid): ez_irgdint_3_nw.inc:20-168, whose statement functions are REAL (each result rounded) */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BATCH ? STB_WAVES : ST_WAVES, 8))) void k_st(ezhip_pts_plan p, float *__restrict__ zout0, const float *__restrict__ zin0,
                                            const float *__restrict__ xs, const float *__restrict__ ys, const int4 *__restrict__ tiles,
                                            int *__restrict__ special_list, unsigned *__restrict__ special_count,
                                            int nfields, size_t in_stride, size_t out_stride)      /* nfields > 1 (c_ezsint_batch_dev): the fields one after the other per tile -- x, y, zones and the
                                                                                                     * list of special points once per batch; pole values of field f in p.polevals[2 f .. 2 f + 1] */
{
    float *__restrict__ zout = zout0;
    const float *__restrict__ zin = zin0;
    typedef uvt_geom<TW, TH> G;
    constexpr int PPT = G::PPT;
    extern __shared__ __attribute__((aligned(16))) float st_lds[];
    unsigned boff = 0;
    if (p.pv_out) {
        if (blockIdx.x < 2) {      /* (the launch's dynamic LDS holds 2052 floats and more) */
            const float *row = blockIdx.x == 0 ? zin + (size_t)(p.pv_nj - 1) * p.ni : zin;
            const float v = block_poleval(row, p.ni, p.pole_weighted, p.ax, st_lds, 2048);
            if (threadIdx.x == 0) p.pv_out[blockIdx.x] = v;
            if (!NW && !BATCH && p.cspec_inline) {
                /* the set's special points (pole points and polar strips; no re-interpolated ones: zones == 1), known to the host: the northern producer takes the
                 * northern ones with the pole value it has just summed, the southern one the southern -- no launch behind the kernel (ez_corrval.c:62-126) */
                const bool north = blockIdx.x == 0;
                for (int k = (int)threadIdx.x; k < p.cspec_count; k += 256) {
                    const int n = p.cspec_list[k];
                    const float fx = p.cspec_x[k], fy = p.cspec_y[k];
                    const int zone = pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, fx, fy);
                    if (north ? (zone == PZ_POLE_N) : (zone == PZ_POLE_S)) { zout[n] = v; continue; }
                    if (!(north ? (zone == PZ_STRIP_N) : (zone == PZ_STRIP_S))) continue;
                    FieldAcc Z;
                    Z.z = zin; Z.ni = p.ni; Z.j1 = p.j1; Z.j2 = p.j2; Z.pole_n = v; Z.pole_s = v; Z.prow_n = nullptr; Z.prow_s = nullptr;      /* (a strip reaches its own pole only) */
                    const int j1s = north ? p.j2 - 2 : p.j1 - 1;
                    zout[n] = p_irgdint_3_wnnc(Z, fx, fy, p.ax, north ? p.ay4_n : p.ay4_s, p.ni, j1s, j1s + 3, p.wrap);
                }
            }
            return;
        }
        boff = 2;
    }
    /* (round 6, as k_uvt) the tiles of the gathering path first: the launch's first uvt_nhb blocks take them from the list behind the table, the blocks in table order skip them */
    const unsigned nhb = (unsigned)p.uvt_nhb, bt = blockIdx.x - boff;
    const bool hb_block = bt < nhb;
    const unsigned b = hb_block ? ((const unsigned *)(tiles + (gridDim.x - boff - nhb)))[bt] : bt - nhb;
    const unsigned tpr = ((unsigned)p.tile_ni + TW - 1u) / TW, by = b / tpr, bx = b - by * tpr, t = threadIdx.x;
    const unsigned cx = bx * TW + (t % TW), cy0 = by * TH + t / TW;
    const bool okx = cx < (unsigned)p.tile_ni;
    float px[PPT], py[PPT];
    const unsigned n0 = okx && cy0 < (unsigned)p.tile_nj ? cy0 * (unsigned)p.tile_ni + cx : 0u, nstep = (unsigned)G::RSTEP * (unsigned)p.tile_ni;
    if (p.uvt_streams) {
        typedef float f2a __attribute__((ext_vector_type(2)));
        const f2a *S = (const f2a *)p.uvt_streams + (size_t)b * (PPT * 256) + t;
#pragma unroll
        for (int k = 0; k < PPT; k++) { const f2a q = __builtin_nontemporal_load(S + k * 256); px[k] = q.x; py[k] = q.y; }
    } else {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const bool ok = okx && cy0 + (unsigned)(G::RSTEP * k) < (unsigned)p.tile_nj;
            const unsigned n = ok ? n0 + (unsigned)k * nstep : 0u;
            px[k] = xs[n]; py[k] = ys[n];
        }
    }
    const int4 tb = tiles[b];
    if (tb.z <= 0) {           /* handed back (the seam, a window beyond the cap) or without a normal point: the gathering path, point by point */
        if (nhb != 0u && !hb_block) return;
#pragma unroll 1
        for (int f = 0; f < (BATCH ? nfields : 1); f++) {
#pragma unroll ST_HB_UNROLL
            for (int k = 0; k < PPT; k++) {
                const unsigned cy = cy0 + (unsigned)(G::RSTEP * k);
                if (okx && cy < (unsigned)p.tile_nj) {
                    const int n = (int)(cy * (unsigned)p.tile_ni + cx);
                    pts1_point<NW ? PK_IRGD3_NW : PK_IRGD3_W>(p, zout0 + (size_t)f * out_stride, zin0 + (size_t)f * in_stride, xs[n], ys[n], n, f == 0 ? special_list : nullptr, special_count, -1, p.polevals + 2 * f);
                }
            }
        }
        return;
    }
    const int i0 = tb.x, j0 = tb.y, W = tb.z, H = tb.w & UVT_H_MASK, ncell = W * H;
    typedef double d2 __attribute__((ext_vector_type(2)));
    float *cells = st_lds;
    d2 *xr = (d2 *)(st_lds + ((ncell + 3) & ~3));
    const int nxr = (W - 3) * 5, nyr = (H - 3) * 5;
    d2 *yr = xr + nxr;
    {
        const d2 *gx = (const d2 *)p.xrec10 + (size_t)i0 * 5, *gy = (const d2 *)p.yrec10 + (size_t)(j0 + 1 - p.j1) * 5;
        for (int idx = (int)t; idx < nxr; idx += 256) xr[idx] = gx[idx];
        for (int idx = (int)t; idx < nyr; idx += 256) yr[idx] = gy[idx];
    }
    /* a batch: the window of field f + 1 is on its way into registers while field f is evaluated (ten cells per thread cover the default 2560 of a tile; a set
     * built under a larger cap stages in place) */
    constexpr int PF = BATCH ? 10 : 1;
    const bool prefetch = BATCH && nfields > 1 && ncell <= PF * 256;
    const unsigned magic = 0xFFFFFFFFu / (unsigned)W + 1u;
    const size_t win0 = (size_t)(j0 - p.j1) * (size_t)p.ni + (size_t)(i0 - 1);
    float nxt[PF];
    const int nf = BATCH ? nfields : 1;                               /* (the one-field instantiation has no loop: 72 VGPRs, seven waves per SIMD) */
#pragma unroll 1
    for (int f = 0; f < nf; f++) {
    zin = zin0 + (size_t)f * in_stride; zout = zout0 + (size_t)f * out_stride;
    const float *pv = p.polevals + 2 * f;
    /* (the per-point state -- zone, indices, records, x, y -- is loop-invariant and the compiler keeps it in registers across the fields: 125 VGPRs, four waves per
     * SIMD, 50.5 us per field of a 16-field cfg3 batch; formed again for every field (113 VGPRs, the same four waves): 55.8) */
    if (f == 0 || !prefetch) {
        if (f) __syncthreads();                                       /* (the cells of the field before are read no more) */
        const float *s1 = zin + win0;
#pragma unroll 4
        for (int idx = (int)t; idx < ncell; idx += 256) {
            const unsigned r = __umulhi((unsigned)idx, magic), c = (unsigned)idx - r * (unsigned)W;
            cells[idx] = s1[(size_t)r * (size_t)p.ni + c];
        }
    } else {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PF; u++) if ((int)(t + 256u * (unsigned)u) < ncell) cells[t + 256u * (unsigned)u] = nxt[u];
    }
    if (prefetch && f + 1 < nfields) {
        const float *s2 = zin + in_stride + win0;
#pragma unroll
        for (int u = 0; u < PF; u++) { const unsigned idx = t + 256u * (unsigned)u, r = __umulhi(idx, magic); nxt[u] = s2[(int)idx < ncell ? r * (unsigned)p.ni + (idx - r * (unsigned)W) : 0u]; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        if (!(okx && cy0 + (unsigned)(G::RSTEP * k) < (unsigned)p.tile_nj)) continue;
        const int n = (int)(n0 + (unsigned)k * nstep);
        const int zone = pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, px[k], py[k]);
        if (zone != PZ_NORMAL) {           /* fill value, or a point of the next kernel's list (as k_pts lists them) */
            if (zone == PZ_FILL) zout[n] = *p.fill;
            else if (!p.pv_out && (zone == PZ_POLE_S || zone == PZ_POLE_N)) zout[n] = pv[zone == PZ_POLE_S ? 1 : 0];
            else if (f == 0 && special_list) {      /* (nullptr: the set's special points are known and ride in the producer blocks) */
                const unsigned long long m = __ballot(1);
                const int lane = (int)__lane_id(), leader = __ffsll((long long)m) - 1;
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(special_count, (unsigned)__popcll(m));
                base = (unsigned)__shfl((int)base, leader, 64);
                special_list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = n;
            }
            continue;
        }
        const int i = min(p.ni - 2 + p.wrap, max(1, max(2 - p.wrap, (int)px[k]))), j = min(p.j2 - 2, max(p.j1 + 1, (int)py[k]));
        /* records {x1, x2 | x3, c1 | c2, c3 | c4, c5 | c6, c5 + c2} (REAL entries widened once per grid) */
        const d2 *xq = xr + (i - 1 - i0) * 5, *yq = yr + (j - 1 - j0) * 5;
        const d2 xa = xq[0], xb = xq[1], xc = xq[2], xd = xq[3], xe = xq[4];
        const float fx2 = (float)xa.y, fx3 = (float)xb.x;                 /* (exact: they were REAL) */
        const double x = (double)(fx2 + (fx3 - fx2) * (px[k] - (float)i));
        const float *cp = cells + (j - 1 - j0) * W + (i - 1 - i0);
        double bb[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double z1 = (double)cp[r * W], z2 = (double)cp[r * W + 1], z3 = (double)cp[r * W + 2], z4 = (double)cp[r * W + 3];
#define ST_RF(e) (NW ? (double)(float)(e) : (e))
            const double a2 = ST_RF(d_fa2(xb.y, z1, z2));
            const double a3 = ST_RF(d_fa3(xb.y, xc.x, xc.y, z1, z2, z3));
            const double a4 = ST_RF(d_fa4(xb.y, xc.x, xc.y, xd.x, xd.y, xe.x, z1, z2, z3, z4));
            bb[r] = ST_RF(d_fa(z1, a2, a3, a4, x, xa.x, xa.y, xb.x));
        }
        const d2 ya = yq[0], yb = yq[1], yc = yq[2], yd = yq[3], ye = yq[4];
        const float fy2 = (float)ya.y, fy3 = (float)yb.x;
        const double y = (double)(fy2 + (fy3 - fy2) * (py[k] - (float)j));
        const double b12 = ST_RF(d_fa2(yb.y, bb[0], bb[1]));
        const double b13 = ST_RF(d_fa3(yb.y, yc.x, yc.y, bb[0], bb[1], bb[2]));
        const double b14 = ST_RF(d_fa4(yb.y, yc.x, yc.y, yd.x, yd.y, ye.x, bb[0], bb[1], bb[2], bb[3]));
#undef ST_RF
        zout[n] = (float)d_fa(bb[0], b12, b13, b14, y, ya.x, ya.y, yb.x);
    }
    }
}

/* ---- k_st1: the bilinear member of the family (ez_irgdint_1_w.inc / ez_irgdint_1_nw.inc from an irregular source) ---------------------------------------------
 * The gathering kernel k_pts<PK_IRGD1_*> takes 54.7 us per 4000 x 2000 field from a rotated 2560 x 1280 source -- as long as the staged bicubic: four gathers of the
 * field and four of the axes per point, no arithmetic to speak of.  Here the tile's window (columns i .. i + 1, rows j .. j + 1 of its normal points) and the axis
 * entries under it are staged as floats; the arithmetic is p_irgdint_1_w's, operand for operand (NW: the clamps of p_irgdint_1_nw).  Tiles on the seam (the wrapped
 * last column has its own x2) are handed back. */
#ifndef ST1_WAVES
#define ST1_WAVES 7
#endif
#define ST1_AXMAX 1024                                 /* axis entries (columns + rows) a tile may stage */
template <bool NW> __device__ __forceinline__ void st1_ij(const ezhip_pts_plan &p, float px, float py, int &i, int &j)
{
    if (NW) { i = min(p.ni - 1, max(1, (int)px)); j = min(p.nj - 1, max(1, (int)py)); }
    else { i = min(p.ni - 2 + p.wrap, max(1, (int)px)); j = min(p.j2 - 1, max(p.j1 + 1, (int)py)); }
}
template <bool NW>
__global__ __launch_bounds__(256) void k_st1_bbox(ezhip_pts_plan p, const float *__restrict__ xs, const float *__restrict__ ys, int4 *__restrict__ tiles, int cap)
{
    typedef uvt_geom<32, 32> G;
    __shared__ int red[4][6];
    const unsigned tpr = ((unsigned)p.tile_ni + 31u) / 32u, b = blockIdx.x, by = b / tpr, bx = b - by * tpr, t = threadIdx.x;
    const unsigned cx = bx * 32 + (t % 32);
    int imin = 0x7fffffff, imax = -1, jmin = 0x7fffffff, jmax = -1, seam = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < G::PPT; k++) {
        const unsigned cy = by * 32 + t / 32 + (unsigned)(G::RSTEP * k);
        if (cx >= (unsigned)p.tile_ni || cy >= (unsigned)p.tile_nj) continue;
        const size_t n = (size_t)cy * p.tile_ni + cx;
        const float px = xs[n], py = ys[n];
        const int zone = pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, px, py);
        if (zone == PZ_FILL) mine = 1;
        if (zone != PZ_NORMAL) continue;
        mine = 1;
        int i, j;
        st1_ij<NW>(p, px, py, i, j);
        if (!NW && ((p.wrap > 0 && i == p.ni - 2 + p.wrap) || j < 0 || i + 1 > p.ni)) seam = 1;
        imin = min(imin, i); imax = max(imax, i); jmin = min(jmin, j); jmax = max(jmax, j);
    }
    imin = uvt_wave_min(imin); jmin = uvt_wave_min(jmin); imax = uvt_wave_max(imax); jmax = uvt_wave_max(jmax); seam = uvt_wave_max(seam); mine = uvt_wave_max(mine);
    if ((t & 63u) == 0) { int *r = red[t >> 6]; r[0] = imin; r[1] = imax; r[2] = jmin; r[3] = jmax; r[4] = seam; r[5] = mine; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < 4; w++) { imin = min(imin, red[w][0]); imax = max(imax, red[w][1]); jmin = min(jmin, red[w][2]); jmax = max(jmax, red[w][3]); seam |= red[w][4]; mine |= red[w][5]; }
        int4 o;
        if (!mine) o = make_int4(0, 0, -1, 0);
        else if (imax < 0) o = make_int4(0, 0, 0, 0);
        else {
            const int W = imax - imin + 2, H = jmax - jmin + 2;
            o = (!seam && W * H <= cap && W + H <= ST1_AXMAX) ? make_int4(imin, jmin, W, H) : make_int4(0, 0, 0, 0);
        }
        tiles[b] = o;
    }
}
/* thread t of a 32 x 32 tile takes FOUR CONSECUTIVE COLUMNS of one row (columns 4 (t % 8) .., row t / 8): its x, y arrive as two 16-byte loads from the set's tile-ordered
 * copy ([tile][half][thread] float4 {x, y, x, y}: a wave's load is one contiguous KB) and its results leave as one 16-byte store.  (Measured against k_st's map -- a
 * column of four rows per thread, 8-byte loads, 4-byte stores: 47.9 against 47.4 us per cfg3 field: the kernel is not bound by the width of its accesses: SQ counters say 116 VALU
 * instructions per point -- two REAL*8 divisions, the zone test in REAL*8, clamps, conversions -- and the SIMDs ~68 % busy; the rest is a block's dependent round trips.) */
__global__ __launch_bounds__(256) void k_st1_pack(ezhip_pts_plan p, const float *__restrict__ xs, const float *__restrict__ ys, float4 *__restrict__ streams)
{
    const unsigned tpr = ((unsigned)p.tile_ni + 31u) / 32u, b = blockIdx.x, by = b / tpr, bx = b - by * tpr, t = threadIdx.x;
    const unsigned cx0 = bx * 32 + (t % 8) * 4, cy = by * 32 + t / 8;
    float v[8];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        v[2 * u] = 0.f; v[2 * u + 1] = 0.f;
        if (cx0 + u < (unsigned)p.tile_ni && cy < (unsigned)p.tile_nj) { const size_t n = (size_t)cy * p.tile_ni + cx0 + u; v[2 * u] = xs[n]; v[2 * u + 1] = ys[n]; }
    }
    streams[((size_t)b * 2 + 0) * 256 + t] = make_float4(v[0], v[1], v[2], v[3]);
    streams[((size_t)b * 2 + 1) * 256 + t] = make_float4(v[4], v[5], v[6], v[7]);
}
template <bool NW, bool BATCH>      /* BATCH: the loop over the fields of a batch (its per-point state lives across the loop: 109 VGPRs; one field: no loop) */
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BATCH ? 4 : ST1_WAVES, 8))) void k_st1(ezhip_pts_plan p, float *__restrict__ zout0, const float *__restrict__ zin0,
                                             const float *__restrict__ xs, const float *__restrict__ ys, const int4 *__restrict__ tiles,
                                             int *__restrict__ special_list, unsigned *__restrict__ special_count, int nfields, size_t in_stride, size_t out_stride)
{
    typedef uvt_geom<32, 32> G;
    constexpr int PPT = G::PPT;
    constexpr int KIND = NW ? PK_IRGD1_NW : PK_IRGD1_W;
    extern __shared__ __attribute__((aligned(16))) float st_lds[];
    const int nf = BATCH ? nfields : 1;
    unsigned boff = 0;
    if (p.pv_out) {
        if (blockIdx.x < 2) {
            const float *row = blockIdx.x == 0 ? zin0 + (size_t)(p.pv_nj - 1) * p.ni : zin0;
            const float v = block_poleval(row, p.ni, p.pole_weighted, p.ax, st_lds, 2048);
            if (threadIdx.x == 0) p.pv_out[blockIdx.x] = v;
            return;
        }
        boff = 2;
    }
    const unsigned tpr = ((unsigned)p.tile_ni + 31u) / 32u, b = blockIdx.x - boff, by = b / tpr, bx = b - by * tpr, t = threadIdx.x;
    const unsigned cx0 = bx * 32 + (t % 8) * 4, cy = by * 32 + t / 8;      /* four consecutive columns of one row */
    const bool oky = cy < (unsigned)p.tile_nj;
    float px[PPT], py[PPT];
    const unsigned n0 = oky && cx0 < (unsigned)p.tile_ni ? cy * (unsigned)p.tile_ni + cx0 : 0u;
    if (p.uvt_streams) {
        typedef float f4a __attribute__((ext_vector_type(4)));
        const f4a *S = (const f4a *)p.uvt_streams + (size_t)b * 512 + t;
        const f4a q0 = __builtin_nontemporal_load(S), q1 = __builtin_nontemporal_load(S + 256);
        px[0] = q0.x; py[0] = q0.y; px[1] = q0.z; py[1] = q0.w; px[2] = q1.x; py[2] = q1.y; px[3] = q1.z; py[3] = q1.w;
    } else {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const bool ok = oky && cx0 + (unsigned)k < (unsigned)p.tile_ni;
            const unsigned n = ok ? n0 + (unsigned)k : 0u;
            px[k] = xs[n]; py[k] = ys[n];
        }
    }
    const int4 tb = tiles[b];
    if (tb.z <= 0) {
#pragma unroll 1
        for (int f = 0; f < nf; f++) {
#pragma unroll 1
            for (int k = 0; k < PPT; k++) {
                if (oky && cx0 + (unsigned)k < (unsigned)p.tile_ni) {
                    const int n = (int)(n0 + (unsigned)k);
                    pts1_point<KIND>(p, zout0 + (size_t)f * out_stride, zin0 + (size_t)f * in_stride, xs[n], ys[n], n, f == 0 ? special_list : nullptr, special_count, -1, p.polevals + 2 * f);
                }
            }
        }
        return;
    }
    const int i0 = tb.x, j0 = tb.y, W = tb.z, H = tb.w, ncell = W * H;
    float *cells = st_lds, *axf = st_lds + ((ncell + 3) & ~3), *ayf = axf + W;
    for (int k = (int)t; k < W; k += 256) axf[k] = p.ax[i0 - 1 + k];
    for (int k = (int)t; k < H; k += 256) ayf[k] = p.ay[j0 - p.j1 + k];
    const unsigned magic = 0xFFFFFFFFu / (unsigned)W + 1u;
    const size_t win0 = (size_t)(j0 - p.j1) * (size_t)p.ni + (size_t)(i0 - 1);
#pragma unroll 1
    for (int f = 0; f < nf; f++) {
        const float *zin = zin0 + (size_t)f * in_stride;
        float *zout = zout0 + (size_t)f * out_stride;
        const float *pv = p.polevals + 2 * f;
        if (f) __syncthreads();
        {
            const float *s1 = zin + win0;
#pragma unroll 4
            for (int idx = (int)t; idx < ncell; idx += 256) {
                const unsigned r = __umulhi((unsigned)idx, magic), c = (unsigned)idx - r * (unsigned)W;
                cells[idx] = s1[(size_t)r * (size_t)p.ni + c];
            }
        }
        __syncthreads();
        float res[PPT];
        unsigned written = 0;                                         /* bit k: this kernel has the value of column k */
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            res[k] = 0.f;
            if (!(oky && cx0 + (unsigned)k < (unsigned)p.tile_ni)) continue;
            const int n = (int)(n0 + (unsigned)k);
            const int zone = pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, px[k], py[k]);
            if (zone != PZ_NORMAL) {
                if (zone == PZ_FILL) { res[k] = *p.fill; written |= 1u << k; }
                else if (!p.pv_out && (zone == PZ_POLE_S || zone == PZ_POLE_N)) { res[k] = pv[zone == PZ_POLE_S ? 1 : 0]; written |= 1u << k; }
                else if (f == 0) {
                    const unsigned long long m = __ballot(1);
                    const int lane = (int)__lane_id(), leader = __ffsll((long long)m) - 1;
                    unsigned base = 0;
                    if (lane == leader) base = atomicAdd(special_count, (unsigned)__popcll(m));
                    base = (unsigned)__shfl((int)base, leader, 64);
                    special_list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = n;
                }
                continue;
            }
            int i, j;
            st1_ij<NW>(p, px[k], py[k], i, j);
            const double x1 = (double)axf[i - i0], x2 = (double)axf[i + 1 - i0];
            const float ayj = ayf[j - j0], ayj1 = ayf[j + 1 - j0];
            const double x = x1 + (x2 - x1) * (double)(px[k] - (float)i);
            const double y = (double)(ayj + (ayj1 - ayj) * (py[k] - (float)j));
            const double dx = (x - x1) / (x2 - x1);
            const double dy = (y - (double)ayj) / (double)(ayj1 - ayj);
            const float *cp = cells + (j - j0) * W + (i - i0);
            const double y1 = d_zlin((double)cp[0], (double)cp[1], dx);
            const double y2 = d_zlin((double)cp[W], (double)cp[W + 1], dx);
            res[k] = (float)d_zlin(y1, y2, dy); written |= 1u << k;
        }
        if (written == 15u && ((p.tile_ni & 3) == 0) && (((size_t)zout & 15) == 0)) {      /* the row pieces of a tile start at multiples of 32 columns: 16-byte aligned when ni is a multiple of 4 and the caller's array is */
            typedef float f4a __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(f4a{res[0], res[1], res[2], res[3]}, (f4a *)(zout + n0));
        } else {
#pragma unroll
            for (int k = 0; k < PPT; k++) if (written >> k & 1u) zout[n0 + (unsigned)k] = res[k];
        }
    }
}

/* (k_uvt as a pipeline -- persistent blocks with two staging buffers, the next tile's window and streams in flight while a tile is computed -- was built twice
 * and measured slower both times: with LDS-DMA staging (the window as separate u / v planes: twice the LDS read instructions) 111 us per cfg3 pair against 90;
 * with the next window held in registers (135 - 155 VGPRs, three waves per SIMD) 96 - 104 us for the kernel against 76.  profiles/r04_experiments.txt.) */
/* The out-of-line strip / re-interpolation code takes the plan BY REFERENCE: a kernel that hands its by-value argument on copies all 400 bytes of it into every
 * thread's scratch first (24 scratch_store_dwordx4 per thread: k_pts_special's 256 blocks wrote 25 MB per launch, half its 10 us).  The plan already lies in
 * memory: the kernel-argument segment, first argument at offset 0 */
#define PLAN_IN_KERNARG() (*(const ezhip_pts_plan *)__builtin_amdgcn_kernarg_segment_ptr())
/* Special points only (a fraction of a percent of a global target): polar strips on the virtual 4-row strip,
 * extrapolation points re-interpolated with degre_extrap. */
__global__ __launch_bounds__(256) void k_pts_special(ezhip_pts_plan p_arg, float *__restrict__ zout, const float *__restrict__ zin,
                                                     const float *__restrict__ xs, const float *__restrict__ ys,
                                                     const int *__restrict__ special_list, const unsigned *__restrict__ special_count,
                                                     unsigned *__restrict__ next_count, int nfields = 1, size_t in_stride = 0, size_t out_stride = 0)
{
    const ezhip_pts_plan &p = PLAN_IN_KERNARG();
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_count = 0;          /* the other counter of the pair: the next launch's */
    const unsigned cnt = *special_count;
    const unsigned total = cnt * (unsigned)nfields;                    /* (a batch: every listed point once per field, pole values of field f in polevals[2 f ..]) */
    for (unsigned kk = blockIdx.x * 256 + threadIdx.x; kk < total; kk += gridDim.x * 256) {
        const unsigned f = kk / cnt, k = kk - f * cnt;
        const int n = special_list[k];
        const float px = xs[n], py = ys[n];
        const size_t o = (p.out_idx ? (size_t)p.out_idx[n] : (size_t)n) + (size_t)f * out_stride;
        const float *pv = p.polevals + 2 * f;
        const int zone = pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, px, py);
        FieldAcc Z;
        Z.z = zin + (size_t)f * in_stride; Z.ni = p.ni; Z.j1 = p.j1; Z.j2 = p.j2;
        Z.pole_n = 0.f; Z.pole_s = 0.f; Z.prow_n = nullptr; Z.prow_s = nullptr;
        if (zone == PZ_REINTERP) { zout[o] = gdinterp_point(p, Z, p.degre_extrap, px, py); continue; }
        if (zone == PZ_POLE_S || zone == PZ_POLE_N) { zout[o] = pv[zone == PZ_POLE_S ? 1 : 0]; continue; }      /* (listed when the pole values came from the launch in front: pv_out) */
        if (p.vector_mode) { Z.prow_n = p.pole_row_n; Z.prow_s = p.pole_row_s; }
        else { Z.pole_n = pv[0]; Z.pole_s = pv[1]; }
        if (p.degree == 3 && p.irregular) {      /* (inline, as in special2c_body) */
            const int north = zone == PZ_STRIP_N, j1s = north ? p.j2 - 2 : p.j1 - 1;
            zout[o] = p_irgdint_3_wnnc(Z, px, py, p.ax, north ? p.ay4_n : p.ay4_s, p.ni, j1s, j1s + 3, p.wrap);
        } else zout[o] = strip_point(p, Z, zone == PZ_STRIP_N, px, py);
    }
}

/* the special points of a wind pair: both components, then the pair's wind matrix, in one launch (two k_pts_special launches and a
 * list pass cost 2 x 8 + 4 us per cfg3 pair, each bound by the latency of its few dependent gathers) */
__global__ __launch_bounds__(256) void k_pts_special2(ezhip_pts_plan p_arg, float *__restrict__ zout1, float *__restrict__ zout2,
                                                      const float *__restrict__ zin1, const float *__restrict__ zin2,
                                                      const float *__restrict__ prow_n2, const float *__restrict__ prow_s2,
                                                      const float *__restrict__ xs, const float *__restrict__ ys,
                                                      const int *__restrict__ special_list, const unsigned *__restrict__ special_count,
                                                      unsigned *__restrict__ next_count)
{
    const ezhip_pts_plan &p = PLAN_IN_KERNARG();
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_count = 0;
    const unsigned cnt = *special_count;
    for (unsigned k = blockIdx.x * 256 + threadIdx.x; k < cnt; k += gridDim.x * 256) {
        const int n = special_list[k];
        const float px = xs[n], py = ys[n];
        const size_t o = p.out_idx ? (size_t)p.out_idx[n] : (size_t)n;
        const int zone = pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, px, py);
        FieldAcc Z1, Z2;
        Z1.z = zin1; Z1.ni = p.ni; Z1.j1 = p.j1; Z1.j2 = p.j2; Z1.pole_n = 0.f; Z1.pole_s = 0.f; Z1.prow_n = nullptr; Z1.prow_s = nullptr;
        Z2 = Z1; Z2.z = zin2;
        float a, b;
        if (zone == PZ_REINTERP) { a = gdinterp_point(p, Z1, p.degre_extrap, px, py); b = gdinterp_point(p, Z2, p.degre_extrap, px, py); }
        else {
            Z1.prow_n = p.pole_row_n; Z1.prow_s = p.pole_row_s; Z2.prow_n = prow_n2; Z2.prow_s = prow_s2;      /* vector mode: synthetic polar wind rows */
            a = strip_point(p, Z1, zone == PZ_STRIP_N, px, py); b = strip_point(p, Z2, zone == PZ_STRIP_N, px, py);
        }
        if (p.wind_M) {
            wm_f2 lo, hi;
            wind_m_load(p.wind_M, p.wind_M_half, o, lo, hi);
            const float u = a, v = b;
            wind_m_apply(lo, hi, p.wind_M_half, u, v, p.wind_dst_rot, a, b);
        }
        zout1[o] = a; zout2[o] = b;
    }
}

/* the same with the points known to the host (ezhip_pts_plan.cspec_*): index, x and y of point k side by side, their number by value.
 * STRIP3: polar-strip points of a bicubic irregular source only (no re-interpolated points in the set: zones == 1) -- nothing but the inline strip, no out-of-line
 * callee: what k_uvt's producer blocks can carry within that kernel's registers */
template <bool STRIP3>
__device__ __forceinline__ void special2c_body(const ezhip_pts_plan &p, float *__restrict__ zout1, float *__restrict__ zout2,
                                               const float *__restrict__ zin1, const float *__restrict__ zin2,
                                               const float *__restrict__ prow_n2, const float *__restrict__ prow_s2, unsigned blk, unsigned nblk, size_t roff /* a batch: the pair's polar wind rows */,
                                               int part /* 0: every point; 1: everything but the southern strip, 2: the southern strip only (the two polar-wind producer blocks of k_uvt: each has its own rows) */)
{
    /* a LANE PAIR per point: lane 2 k takes the first component, lane 2 k + 1 the second (the few special points of a set are a chain of dependent gathers --
     * ~12 us of latency per call with both components one after the other on one lane), then they swap results for the wind matrix */
    const unsigned cnt2 = 2u * (unsigned)p.cspec_count;
    for (unsigned k2 = blk * 256 + threadIdx.x; k2 < cnt2; k2 += nblk * 256) {
        const unsigned k = k2 >> 1, comp = k2 & 1u;
        const int n = p.cspec_list[k];
        const float px = p.cspec_x[k], py = p.cspec_y[k];
        const size_t o = p.out_idx ? (size_t)p.out_idx[n] : (size_t)n;
        wm_f2 mlo = {1.f, 0.f}, mhi = {0.f, 1.f};
        if (p.wind_M) wind_m_load(p.wind_M, p.wind_M_half, o, mlo, mhi);       /* on its way while the stencils are gathered */
        const int zone = pts_zone(p.zones, p.ni, p.nj, p.j1, p.j2, p.ypole_n, p.ypole_s, p.vector_mode, p.degre_extrap, px, py);
        if (part && (part == 2) != (zone == PZ_STRIP_S)) continue;      /* (both lanes of a pair alike) */
        FieldAcc Z;
        Z.z = comp ? zin2 : zin1; Z.ni = p.ni; Z.j1 = p.j1; Z.j2 = p.j2; Z.pole_n = 0.f; Z.pole_s = 0.f; Z.prow_n = nullptr; Z.prow_s = nullptr;
        float mine;
        if (!STRIP3 && zone == PZ_REINTERP) mine = gdinterp_point(p, Z, p.degre_extrap, px, py);
        else {
            Z.prow_n = comp ? prow_n2 : p.pole_row_n; Z.prow_s = comp ? prow_s2 : p.pole_row_s;
            if (roff) { if (Z.prow_n) Z.prow_n += roff; if (Z.prow_s) Z.prow_s += roff; }
            /* (the bicubic strip of an irregular source -- cfg3's case -- inline: through the out-of-line strip_point this launch of ONE thread block spent most of
             * its 10 us fetching instructions and passing arguments through scratch) */
            if (STRIP3 || (p.degree == 3 && p.irregular)) {
                const int north = zone == PZ_STRIP_N, j1s = north ? p.j2 - 2 : p.j1 - 1;
                mine = p_irgdint_3_wnnc(Z, px, py, p.ax, north ? p.ay4_n : p.ay4_s, p.ni, j1s, j1s + 3, p.wrap);
            } else mine = strip_point(p, Z, zone == PZ_STRIP_N, px, py);
        }
        const float other = __shfl_xor(mine, 1, 64);                           /* (both lanes of a pair are in the loop together: cnt2 is even, the stride too) */
        float a = comp ? other : mine, b = comp ? mine : other;
        if (p.wind_M) { const float u = a, v = b; wind_m_apply(mlo, mhi, p.wind_M_half, u, v, p.wind_dst_rot, a, b); }
        if (comp) zout2[o] = b; else zout1[o] = a;
    }
}
__global__ __launch_bounds__(256) void k_pts_special2c(ezhip_pts_plan p_arg, float *__restrict__ zout1, float *__restrict__ zout2,
                                                       const float *__restrict__ zin1, const float *__restrict__ zin2,
                                                       const float *__restrict__ prow_n2, const float *__restrict__ prow_s2)
{
    const ezhip_pts_plan &p = PLAN_IN_KERNARG();
    const size_t f = blockIdx.y;                                                  /* the pair of a batch (c_ezuvint_batch_dev); 0 otherwise */
    special2c_body<false>(p, zout1 + f * p.pair_out_stride, zout2 + f * p.pair_out_stride, zin1 + f * p.pair_in_stride, zin2 + f * p.pair_in_stride, prow_n2, prow_s2, blockIdx.x, gridDim.x,
                   f * (size_t)p.pair_rows_stride);
}
__global__ __launch_bounds__(256) void k_spec_gather(int *__restrict__ list_out, float *__restrict__ x_out, float *__restrict__ y_out,
                                                     const int *__restrict__ list_in, const float *__restrict__ xs, const float *__restrict__ ys, unsigned cnt)
{
    const unsigned k = blockIdx.x * 256 + threadIdx.x;
    if (k >= cnt) return;
    const int n = list_in[k];
    list_out[k] = n; x_out[k] = xs[n]; y_out[k] = ys[n];
}

/* per host thread: the list of special point indices of a launch and a PAIR of counters (the special kernel of
 * launch e consumes counter e & 1 and zeroes the other one for launch e + 1: no memset between launches) */
static thread_local struct { int *list; unsigned *count; size_t cap; unsigned epoch; } t_spec;
static void kernels_thread_release(void)
{
    if (t_spec.list) { (void)hipFree(t_spec.list); t_spec.list = nullptr; t_spec.count = nullptr; t_spec.cap = 0; }
    for (int k = 0; k < 2; k++) {
        if (t_bnc.buf[k]) { (void)hipHostFree(t_bnc.buf[k]); t_bnc.buf[k] = nullptr; }
        if (t_bnc.ev[k]) { (void)hipEventDestroy(t_bnc.ev[k]); t_bnc.ev[k] = nullptr; }
    }
    t_bnc.ok = false;
    if (t_side) { (void)hipStreamDestroy(t_side); t_side = nullptr; }
    if (t_ev_fork) { (void)hipEventDestroy(t_ev_fork); t_ev_fork = nullptr; }
    if (t_ev_join) { (void)hipEventDestroy(t_ev_join); t_ev_join = nullptr; }
    t_side_pending = false;
}

static int interp_pts_impl(const ezhip_pts_plan *plan, float *d_zout, const float *d_zin, const float *d_x, const float *d_y, int npts, int nfields, size_t in_stride, size_t out_stride);
extern "C" int ezhip_interp_pts(const ezhip_pts_plan *plan, float *d_zout, const float *d_zin,
                                const float *d_x, const float *d_y, int npts)
{
    return interp_pts_impl(plan, d_zout, d_zin, d_x, d_y, npts, 1, 0, 0);
}
/* nfields fields of one grid set through the staged-tile kernel in ONE launch (plan->uvt_tiles set, plan->polevals: 2 values per field, no pv_out): -2 when the
 * plan is not on that path */
extern "C" int ezhip_interp_pts_batch(const ezhip_pts_plan *plan, float *d_zout, const float *d_zin, const float *d_x, const float *d_y, int npts,
                                      int nfields, size_t in_stride, size_t out_stride)
{
    const int kind = pts_kind(plan);
    const bool cubic = (kind == PK_IRGD3_W || (kind == PK_IRGD3_NW && plan->i1 == 1 && plan->i2 == plan->ni)) && plan->xrec10, linear = kind == PK_IRGD1_W || kind == PK_IRGD1_NW;
    if (!plan->uvt_tiles || plan->pv_out || !(cubic || linear) || plan->tile_ni <= 0 || plan->out_idx || nfields < 1) return -2;
    return interp_pts_impl(plan, d_zout, d_zin, d_x, d_y, npts, nfields, in_stride, out_stride);
}
static int interp_pts_impl(const ezhip_pts_plan *plan, float *d_zout, const float *d_zin, const float *d_x, const float *d_y, int npts, int nfields, size_t in_stride, size_t out_stride)
{
    if (npts <= 0) return 0;
    const dim3 grid((npts + 255) / 256 + (plan->pv_out ? 2 : 0)), block(256);
    if (t_spec.cap < (size_t)npts) {
        (void)hipStreamSynchronize(g_stream);
        if (t_spec.list) (void)hipFree(t_spec.list);
        t_spec.list = nullptr; t_spec.cap = 0;
        if (hipMalloc((void **)&t_spec.list, sizeof(int) * (size_t)npts + 64) != hipSuccess) return set_err(hipGetLastError(), "k_pts special list");
        t_spec.count = (unsigned *)(t_spec.list + npts);
        if (hipMemsetAsync(t_spec.count, 0, 64, g_stream) != hipSuccess) return -1;
        t_spec.cap = (size_t)npts; t_spec.epoch = 0;
    }
    unsigned *cnt = t_spec.count + (t_spec.epoch & 1), *cnt_next = t_spec.count + ((t_spec.epoch + 1) & 1);
    t_spec.epoch++;
    const int kind_st = pts_kind(plan);
    if (plan->uvt_tiles && (kind_st == PK_IRGD1_W || kind_st == PK_IRGD1_NW) && plan->tile_ni > 0 && !plan->out_idx) {
        /* the bilinear staged-tile kernel (the set's table was built for this degree and these zone options) */
        const int cap = plan->uvt_cap > 0 ? plan->uvt_cap : UVT_CAP_DEFAULT;
        const unsigned nt = (unsigned)ezhip_uvt_ntiles(plan, 3232);
        size_t lds = 4 * (size_t)((cap + 3) & ~3) + 4 * (size_t)ST1_AXMAX + 16;
        if (lds < 4 * 2052 + 16) lds = 4 * 2052 + 16;
        if (lds > 65536) return -1;
#define ST1_LAUNCH(NW, B) hipLaunchKernelGGL((k_st1<NW, B>), dim3(nt + (plan->pv_out ? 2u : 0u)), block, lds, g_stream, *plan, d_zout, d_zin, d_x, d_y, (const int4 *)plan->uvt_tiles, t_spec.list, cnt, nfields, in_stride, out_stride)
        if (kind_st == PK_IRGD1_W) { if (nfields > 1) ST1_LAUNCH(false, true); else ST1_LAUNCH(false, false); }
        else { if (nfields > 1) ST1_LAUNCH(true, true); else ST1_LAUNCH(true, false); }
#undef ST1_LAUNCH
        if (LAUNCH_CHECK("k_st1")) return -1;
    } else
    if (plan->uvt_tiles && (kind_st == PK_IRGD3_W || (kind_st == PK_IRGD3_NW && plan->i1 == 1 && plan->i2 == plan->ni)) && plan->tile_ni > 0 && !plan->out_idx && plan->xrec10) {
        /* the scalar staged-tile kernel (the set's table was built under this plan's zone options) */
        const int cap = plan->uvt_cap > 0 ? plan->uvt_cap : UVT_CAP_DEFAULT;
        const unsigned nt = (unsigned)ezhip_uvt_ntiles(plan, 3232);
        size_t lds = 4 * (size_t)((cap + 3) & ~3) + 80 * (size_t)UVT_REC_MAX + 16;
        if (lds < 4 * 2052 + 16) lds = 4 * 2052 + 16;
        if (lds > 65536) return -1;                      /* (no attribute needed up to 64 KB) */
        /* the set's special points known to the host, few, none re-interpolated, and producer blocks in the launch: they take the points along (k_st), nothing is listed,
         * no launch behind the kernel (EZHIP_ST_SPECIAL_LAUNCH=1: as before) */
        if (kind_st == PK_IRGD3_W && nfields == 1 && plan->pv_out && plan->cspec_valid && plan->cspec_count > 0 && plan->cspec_count <= 4096 && plan->zones == 1 && !plan->vector_mode && !getenv("EZHIP_ST_SPECIAL_LAUNCH")) {
            ezhip_pts_plan pi = *plan;
            pi.cspec_inline = 1;
            t_spec.epoch--;                                          /* (no list, no counter used by this launch) */
            hipLaunchKernelGGL((k_st<32, 32, false, false>), dim3(nt + (unsigned)pi.uvt_nhb + 2u), block, lds, g_stream, pi, d_zout, d_zin, d_x, d_y, (const int4 *)plan->uvt_tiles, (int *)nullptr, cnt, 1, in_stride, out_stride);
            return LAUNCH_CHECK("k_st");
        }
#define ST_LAUNCH(NW, B) hipLaunchKernelGGL((k_st<32, 32, NW, B>), dim3(nt + (unsigned)plan->uvt_nhb + (plan->pv_out ? 2u : 0u)), block, lds, g_stream, *plan, d_zout, d_zin, d_x, d_y, (const int4 *)plan->uvt_tiles, t_spec.list, cnt, nfields, in_stride, out_stride)
        if (kind_st == PK_IRGD3_W) { if (nfields > 1) ST_LAUNCH(false, true); else ST_LAUNCH(false, false); }
        else { if (nfields > 1) ST_LAUNCH(true, true); else ST_LAUNCH(true, false); }
#undef ST_LAUNCH
        if (LAUNCH_CHECK("k_st")) return -1;
    } else
#define PTS_CASE(K) case K: hipLaunchKernelGGL(k_pts<K>, grid, block, 0, g_stream, *plan, d_zout, d_zin, d_x, d_y, npts, t_spec.list, cnt); break
    switch (pts_kind(plan)) {
    PTS_CASE(PK_RGD0); PTS_CASE(PK_RGD1_NW); PTS_CASE(PK_RGD1_W); PTS_CASE(PK_RGD3_NW); PTS_CASE(PK_RGD3_W);
    PTS_CASE(PK_IRGD1_NW); PTS_CASE(PK_IRGD1_W); PTS_CASE(PK_IRGD3_NW); PTS_CASE(PK_IRGD3_W);
    }
#undef PTS_CASE
    if (LAUNCH_CHECK("k_pts")) return -1;
    if (ezhip_side_join()) return -1;       /* the special points read the polar wind rows a side stream may still be producing */
    /* always launched (it also re-arms the counter pair); a grid-stride loop over the few listed points */
    hipLaunchKernelGGL(k_pts_special, dim3(npts < 65536 ? 16 : 256), block, 0, g_stream, *plan, d_zout, d_zin, d_x, d_y, t_spec.list, cnt, cnt_next, nfields, in_stride, out_stride);
    return LAUNCH_CHECK("k_pts_special");
}

/* vector pair: plan_u / plan_v differ only in their polar wind rows (read by the special points) */
extern "C" int ezhip_interp_pts2(const ezhip_pts_plan *plan_u, const ezhip_pts_plan *plan_v, float *d_out_u, float *d_out_v,
                                 const float *d_in_u, const float *d_in_v, const float *d_x, const float *d_y, int npts)
{
    if (npts <= 0) return 0;
    const dim3 block(256);
    if (t_spec.cap < (size_t)npts) {
        (void)hipStreamSynchronize(g_stream);
        if (t_spec.list) (void)hipFree(t_spec.list);
        t_spec.list = nullptr; t_spec.cap = 0;
        if (hipMalloc((void **)&t_spec.list, sizeof(int) * (size_t)npts + 64) != hipSuccess) return set_err(hipGetLastError(), "k_pts special list");
        t_spec.count = (unsigned *)(t_spec.list + npts);
        if (hipMemsetAsync(t_spec.count, 0, 64, g_stream) != hipSuccess) return -1;
        t_spec.cap = (size_t)npts; t_spec.epoch = 0;
    }
    const bool cached = plan_u->cspec_valid && plan_u->vector_mode;      /* the set's special points are known: nothing is listed, no counter is used */
    unsigned *cnt = t_spec.count + (t_spec.epoch & 1), *cnt_next = t_spec.count + ((t_spec.epoch + 1) & 1);
    if (!cached) t_spec.epoch++;
    int *list_arg = cached ? nullptr : t_spec.list;
    ezhip_pts_plan pu2 = *plan_u;
    pu2.newton_literal = getenv("EZHIP_WIND_NEWTON_LITERAL") ? 1 : 0;      /* development: the reference's literal Newton form for winds too */
    pu2.xcd_order = getenv("EZHIP_PTS_XCD") ? 1 : 0;                        /* development: XCD k takes the k-th eighth of the blocks (fewer fabric reads, measured slower) */
    if (getenv("EZHIP_PTS_NOTILE") || (long long)pu2.tile_ni * pu2.tile_nj != (long long)npts || pu2.out_idx) pu2.tile_ni = pu2.tile_nj = 0;
    const bool fast3w = pts_kind(plan_u) == PK_IRGD3_W && !pu2.newton_literal;
    const bool stage3nw = pts_kind(plan_u) == PK_IRGD3_NW && plan_u->i1 == 1 && plan_u->i2 == plan_u->ni;      /* a regional source: k_uvt's literal twin */
    const int npairs = pu2.npairs > 1 ? pu2.npairs : 1;
    if (npairs > 1 && !(cached && fast3w && pu2.uvt_tiles && pu2.uvt_streams && pu2.tile_ni > 0 && pu2.xrec10 && pu2.yrec10 && !pu2.out_idx && pu2.uvt_shape == 3232
                        && (!pu2.pw_out || !getenv("EZHIP_POLAR_WIND_SIDE")))) return -2;      /* c_ezuvint_batch_dev: only the staged-tile kernel has a batch form (the caller goes pair by pair) */
    if (pu2.pw_out && (!fast3w || getenv("EZHIP_POLAR_WIND_SIDE"))) {
        /* kernels without the producer blocks: the rows come from k_polar_wind on the side stream, joined below before the special points */
        if (ezhip_side_begin()) return -1;
        hipLaunchKernelGGL(k_polar_wind<false>, dim3(2), dim3(1024), 0, g_stream, pu2.pw_out, d_in_u, d_in_v, pu2.pw_plon2, pu2.ni, pu2.nj, pu2.pw_xg4_n, pu2.pw_xg4_s, pu2.pw_weighted, pu2.pw_ax);
        const int bad = LAUNCH_CHECK("k_polar_wind");
        if (ezhip_side_end() || bad) return -1;
        pu2.pw_out = nullptr;
    }
    /* a wave = 16 x 4 target points, a block = 64 x 4 (tools/sweep_cfg3.py, interleaved, cfg3: 108.7 us per pair; a wave 8 x 8 in a block 32 x 8 -- the first tile order --
     * 111.1; 16 x 4 in a block 32 x 8: 109.1; in a block 16 x 16: 114.0; a wave 32 x 2: 117.6; 4 x 16: 136.2).  EZHIP_PTS_TILE: 0 = 8 x 8, 2 .. 5 the others */
    pu2.tile_shape = getenv("EZHIP_PTS_TILE") ? atoi(getenv("EZHIP_PTS_TILE")) : 1;
    const int tbw = pu2.tile_shape == 1 ? 64 : pu2.tile_shape == 2 ? 16 : pu2.tile_shape == 3 ? 128 : pu2.tile_shape == 5 ? 16 : 32;
    const int tbh = pu2.tile_shape == 1 ? 4 : pu2.tile_shape == 2 ? 16 : pu2.tile_shape == 3 ? 2 : pu2.tile_shape == 5 ? 16 : 8;
    const dim3 grid((pu2.tile_ni > 0 ? (unsigned)(((pu2.tile_ni + tbw - 1) / tbw) * ((pu2.tile_nj + tbh - 1) / tbh)) : (unsigned)((npts + 255) / 256)) + (pu2.pw_out ? 2u : 0u));
    if (cached && (fast3w || stage3nw) && pu2.uvt_tiles && pu2.tile_ni > 0 && pu2.xrec10 && pu2.yrec10 && !pu2.out_idx) {
        /* the grid set's tile table is known: stencil windows staged in LDS (k_uvt); the set's special points behind it */
        pu2.uvt_debug = EZH_DEVINT("EZHIP_UVT_DEBUG");
        const unsigned nt = (unsigned)ezhip_uvt_ntiles(&pu2, pu2.uvt_shape);
        const int4 *tl = (const int4 *)pu2.uvt_tiles;
        const dim3 g(nt + (unsigned)pu2.uvt_nhb + (pu2.pw_out ? 2u * (unsigned)npairs : 0u));
        size_t lds = (size_t)8 * (size_t)pu2.uvt_cap + (stage3nw ? 80 : 32) * UVT_REC_MAX;
        if (lds < 4 * 2052 + 16) lds = 4 * 2052 + 16;                          /* (the polar-wind producer blocks' row buffer) */
        pu2.xcd_order = getenv("EZHIP_UVT_XCD") ? atoi(getenv("EZHIP_UVT_XCD")) : 0;
        /* few special points and polar-wind producer blocks in the launch: the producers take them along (EZHIP_UVT_SPECIAL_LAUNCH=1: the launch of their own, as before) */
        pu2.cspec_inline = pu2.pw_out && !stage3nw && fast3w && pu2.zones == 1 && plan_u->cspec_count > 0 && plan_u->cspec_count <= 512 && pu2.pole_row_n == pu2.pw_out && !(pu2.uvt_debug & 8) && !getenv("EZHIP_UVT_SPECIAL_LAUNCH");
        pu2.uvt_read2 = getenv("EZHIP_UVT_READ2") ? 1 : 0;                      /* development: the compiler's paired LDS reads (same results) */
#define UVT_LAUNCH(TW, TH) hipLaunchKernelGGL((k_uvt<TW, TH>), g, block, lds, g_stream, pu2, d_out_u, d_out_v, d_in_u, d_in_v, d_x, d_y, tl)
        if (npairs > 1) hipLaunchKernelGGL((k_uvt<32, 32, false, true>), g, block, lds, g_stream, pu2, d_out_u, d_out_v, d_in_u, d_in_v, d_x, d_y, tl);      /* all pairs of the batch */
        else if (stage3nw) hipLaunchKernelGGL((k_uvt<32, 32, true>), g, block, lds, g_stream, pu2, d_out_u, d_out_v, d_in_u, d_in_v, d_x, d_y, tl);      /* (tables of regional sets are built with 32 x 32 tiles) */
        else
        switch (pu2.uvt_shape) { case 3216: UVT_LAUNCH(32, 16); break; case 6408: UVT_LAUNCH(64, 8); break; case 6416: UVT_LAUNCH(64, 16); break; default: UVT_LAUNCH(32, 32); break; }
#undef UVT_LAUNCH
        if (LAUNCH_CHECK("k_uvt")) return -1;
        if (ezhip_side_join()) return -1;
        if (plan_u->cspec_count > 0 && !(pu2.uvt_debug & 8) && !pu2.cspec_inline) {
            const unsigned nbk = (unsigned)((2 * plan_u->cspec_count + 255) / 256);      /* a lane pair per point */
            hipLaunchKernelGGL(k_pts_special2c, dim3(nbk < 256 ? nbk : 256, (unsigned)npairs), block, 0, g_stream, *plan_u, d_out_u, d_out_v, d_in_u, d_in_v, plan_v->pole_row_n, plan_v->pole_row_s);
            return LAUNCH_CHECK("k_pts_special2c");
        }
        return 0;
    }
#define PTS2_CASE(K) case K: if (K == PK_IRGD3_W && !pu2.newton_literal) hipLaunchKernelGGL(k_pts2_irgd3w, grid, block, 0, g_stream, pu2, d_out_u, d_out_v, d_in_u, d_in_v, d_x, d_y, npts, list_arg, cnt); \
        else hipLaunchKernelGGL(k_pts2<K>, grid, block, 0, g_stream, pu2, d_out_u, d_out_v, d_in_u, d_in_v, d_x, d_y, npts, list_arg, cnt); break
    switch (pts_kind(plan_u)) {
    PTS2_CASE(PK_RGD0); PTS2_CASE(PK_RGD1_NW); PTS2_CASE(PK_RGD1_W); PTS2_CASE(PK_RGD3_NW); PTS2_CASE(PK_RGD3_W);
    PTS2_CASE(PK_IRGD1_NW); PTS2_CASE(PK_IRGD1_W); PTS2_CASE(PK_IRGD3_NW); PTS2_CASE(PK_IRGD3_W);
    }
#undef PTS2_CASE
    if (LAUNCH_CHECK("k_pts2")) return -1;
    if (ezhip_side_join()) return -1;
    if (cached) {
        if (plan_u->cspec_count > 0) {
            const unsigned nbk = (unsigned)((2 * plan_u->cspec_count + 255) / 256);      /* a lane pair per point */
            hipLaunchKernelGGL(k_pts_special2c, dim3(nbk < 256 ? nbk : 256), block, 0, g_stream, *plan_u, d_out_u, d_out_v, d_in_u, d_in_v, plan_v->pole_row_n, plan_v->pole_row_s);
            return LAUNCH_CHECK("k_pts_special2c");
        }
        return 0;
    }
    if (!plan_u->vector_mode) {          /* two scalar fields sharing a point list: their pole values differ, one launch each */
        hipLaunchKernelGGL(k_pts_special, dim3(npts < 65536 ? 16 : 256), block, 0, g_stream, *plan_u, d_out_u, d_in_u, d_x, d_y, t_spec.list, cnt, cnt_next);
        hipLaunchKernelGGL(k_pts_special, dim3(npts < 65536 ? 16 : 256), block, 0, g_stream, *plan_v, d_out_v, d_in_v, d_x, d_y, t_spec.list, cnt, cnt_next);
        return LAUNCH_CHECK("k_pts_special");
    }
    /* the listed points: both components and the wind matrix in one launch (vector mode: the plans differ in their polar wind rows only) */
    hipLaunchKernelGGL(k_pts_special2, dim3(npts < 65536 ? 16 : 256), block, 0, g_stream, *plan_u, d_out_u, d_out_v, d_in_u, d_in_v,
                       plan_v->pole_row_n, plan_v->pole_row_s, d_x, d_y, t_spec.list, cnt, cnt_next);
    return LAUNCH_CHECK("k_pts_special2");
}

/* shape = 100 TW + TH: 3232 (default), 3216, 6416, 6408 */
static void uvt_dims(int shape, int *tw, int *th) { *tw = shape / 100; *th = shape % 100; if (!((*tw == 32 && (*th == 32 || *th == 16)) || (*tw == 64 && (*th == 16 || *th == 8)))) { *tw = 32; *th = 32; } }
extern "C" int ezhip_uvt_ntiles(const ezhip_pts_plan *plan, int shape)
{
    int tw, th;
    uvt_dims(shape, &tw, &th);
    if (plan->tile_ni <= 0 || plan->tile_nj <= 0) return 0;
    return ((plan->tile_ni + tw - 1) / tw) * ((plan->tile_nj + th - 1) / th);
}
extern "C" int ezhip_uvt_build(const ezhip_pts_plan *plan, const float *d_x, const float *d_y, void *d_tiles, int shape, int *stats, int want_list)
{
    const int nt = ezhip_uvt_ntiles(plan, shape);
    if (nt <= 0 || !d_tiles) return -1;
    const int cap = plan->uvt_cap > 0 ? plan->uvt_cap : UVT_CAP_DEFAULT;
    int tw, th;
    uvt_dims(shape, &tw, &th);
    const int recmax = UVT_REC_MAX;
#define UVT_BB(TW, TH) hipLaunchKernelGGL((k_uvt_bbox<TW, TH>), dim3(nt), dim3(256), 0, g_stream, *plan, d_x, d_y, (int4 *)d_tiles, cap, recmax)
    if (tw == 32 && th == 16) UVT_BB(32, 16); else if (tw == 64 && th == 8) UVT_BB(64, 8); else if (tw == 64 && th == 16) UVT_BB(64, 16); else UVT_BB(32, 32);
#undef UVT_BB
    if (LAUNCH_CHECK("k_uvt_bbox")) return -1;
    if (set_err(hipStreamSynchronize(g_stream), "k_uvt_bbox")) return -1;
    if (stats) {
        int4 *h = (int4 *)malloc(sizeof(int4) * (size_t)nt);
        if (!h) return -1;
        if (set_err(hipMemcpy(h, d_tiles, sizeof(int4) * (size_t)nt, hipMemcpyDeviceToHost), "k_uvt_bbox tiles")) { free(h); return -1; }
        stats[0] = stats[1] = stats[2] = stats[3] = 0;
        for (int k = 0; k < nt; k++) { if (h[k].z > 0) { stats[0]++; if (h[k].z * (h[k].w & UVT_H_MASK) > stats[3]) stats[3] = h[k].z * (h[k].w & UVT_H_MASK); } else if (h[k].z == 0) stats[1]++; else stats[2]++; }
        if (want_list && stats[1] + (want_list == 2 ? stats[2] : 0) > 0) {         /* the handed-back tiles' indices behind the table: the launch's first blocks take them (they run several times as long as a staged tile); 2: k_st, whose gathering path also takes the tiles without a normal point */
            unsigned *l = (unsigned *)malloc(sizeof(unsigned) * (size_t)(stats[1] + stats[2]));
            if (!l) { free(h); return -1; }
            int m = 0;
            for (int k = 0; k < nt; k++) if (h[k].z == 0 || (want_list == 2 && h[k].z < 0)) l[m++] = (unsigned)k;
            const int bad = set_err(hipMemcpy((char *)d_tiles + sizeof(int4) * (size_t)nt, l, sizeof(unsigned) * (size_t)m, hipMemcpyHostToDevice), "k_uvt_bbox list");
            free(l);
            if (bad) { free(h); return -1; }
        }
        free(h);
    }
    return 0;
}

/* the tile-ordered stream copy of a wind-pair plan (k_uvt_pack): d_streams holds 12 bytes x 256 x PPT x ntiles (+ 16).  plan->wind_M: NULL or the packed-rotation form */
extern "C" size_t ezhip_uvt_stream_bytes(const ezhip_pts_plan *plan, int shape)
{
    int tw, th;
    uvt_dims(shape, &tw, &th);
    return (size_t)12 * (size_t)tw * (size_t)th * (size_t)ezhip_uvt_ntiles(plan, shape) + 16;
}
extern "C" int ezhip_uvt_pack_streams(const ezhip_pts_plan *plan, const float *d_x, const float *d_y, void *d_streams, int shape)
{
    const int nt = ezhip_uvt_ntiles(plan, shape);
    if (nt <= 0 || !d_streams || (plan->wind_M && !plan->wind_M_half)) return -1;
    int tw, th;
    uvt_dims(shape, &tw, &th);
#define UVT_PK(TW, TH) hipLaunchKernelGGL((k_uvt_pack<TW, TH>), dim3(nt), dim3(256), 0, g_stream, *plan, d_x, d_y, (uvt_rec *)d_streams)
    if (tw == 32 && th == 16) UVT_PK(32, 16); else if (tw == 64 && th == 8) UVT_PK(64, 8); else if (tw == 64 && th == 16) UVT_PK(64, 16); else UVT_PK(32, 32);
#undef UVT_PK
    if (LAUNCH_CHECK("k_uvt_pack")) return -1;
    return set_err(hipStreamSynchronize(g_stream), "k_uvt_pack");
}

/* the tile table of the bilinear kernel (k_st1_bbox): as ezhip_uvt_build, 32 x 32 tiles */
extern "C" int ezhip_st1_build(const ezhip_pts_plan *plan, const float *d_x, const float *d_y, void *d_tiles, int *stats)
{
    const int nt = ezhip_uvt_ntiles(plan, 3232), kind = pts_kind(plan);
    if (nt <= 0 || !d_tiles || !(kind == PK_IRGD1_W || kind == PK_IRGD1_NW)) return -1;
    const int cap = plan->uvt_cap > 0 ? plan->uvt_cap : UVT_CAP_DEFAULT;
    if (kind == PK_IRGD1_W) hipLaunchKernelGGL(k_st1_bbox<false>, dim3(nt), dim3(256), 0, g_stream, *plan, d_x, d_y, (int4 *)d_tiles, cap);
    else hipLaunchKernelGGL(k_st1_bbox<true>, dim3(nt), dim3(256), 0, g_stream, *plan, d_x, d_y, (int4 *)d_tiles, cap);
    if (LAUNCH_CHECK("k_st1_bbox") || set_err(hipStreamSynchronize(g_stream), "k_st1_bbox")) return -1;
    if (stats) {
        int4 *h = (int4 *)malloc(sizeof(int4) * (size_t)nt);
        if (!h) return -1;
        if (set_err(hipMemcpy(h, d_tiles, sizeof(int4) * (size_t)nt, hipMemcpyDeviceToHost), "k_st1_bbox tiles")) { free(h); return -1; }
        stats[0] = stats[1] = stats[2] = stats[3] = 0;
        for (int k = 0; k < nt; k++) { if (h[k].z > 0) { stats[0]++; if (h[k].z * h[k].w > stats[3]) stats[3] = h[k].z * h[k].w; } else if (h[k].z == 0) stats[1]++; else stats[2]++; }
        free(h);
    }
    return 0;
}
extern "C" int ezhip_st_pack_streams(const ezhip_pts_plan *plan, const float *d_x, const float *d_y, void *d_streams)
{
    const int nt = ezhip_uvt_ntiles(plan, 3232), kind = pts_kind(plan);
    if (nt <= 0 || !d_streams) return -1;
    if (kind == PK_IRGD1_W || kind == PK_IRGD1_NW) hipLaunchKernelGGL(k_st1_pack, dim3(nt), dim3(256), 0, g_stream, *plan, d_x, d_y, (float4 *)d_streams);      /* (the bilinear kernel's thread-to-point map) */
    else
    hipLaunchKernelGGL((k_st_pack<32, 32>), dim3(nt), dim3(256), 0, g_stream, *plan, d_x, d_y, (float2 *)d_streams);
    if (LAUNCH_CHECK("k_st_pack")) return -1;
    return set_err(hipStreamSynchronize(g_stream), "k_st_pack");
}

extern "C" int ezhip_pts2_special_snapshot(int *d_list_out, float *d_x_out, float *d_y_out, int cap, const float *d_xs, const float *d_ys)
{
    if (!t_spec.count || t_spec.epoch == 0) return -1;
    unsigned h = 0;
    /* the counter the last launch consumed keeps its value until the launch after the next one zeroes it */
    if (set_err(hipMemcpyAsync(&h, t_spec.count + ((t_spec.epoch - 1) & 1), sizeof(h), hipMemcpyDeviceToHost, g_stream), "special count") ||
        set_err(hipStreamSynchronize(g_stream), "special count")) return -1;
    if (h > (unsigned)0x7fffffff || (size_t)h > t_spec.cap) return -1;
    if (d_list_out && h > 0) {
        if ((unsigned)cap < h) return -1;
        hipLaunchKernelGGL(k_spec_gather, dim3((h + 255) / 256), dim3(256), 0, g_stream, d_list_out, d_x_out, d_y_out, t_spec.list, d_xs, d_ys, h);
        if (LAUNCH_CHECK("k_spec_gather") || set_err(hipStreamSynchronize(g_stream), "special points")) return -1;
    }
    return (int)h;
}

/* ===================================================================================== */
/* small reductions                                                                         */
/* ===================================================================================== */
__global__ __launch_bounds__(256) void k_polevals(float *out2, const float *zin, size_t field_stride, int ni, int nj, int weighted, const float *ax)
{
    __shared__ __attribute__((aligned(16))) float lds4[POLE_CHUNK + 4];
    zin += blockIdx.y * field_stride;
    const float *row = blockIdx.x == 0 ? zin + (size_t)(nj - 1) * ni : zin;
    float v = block_poleval(row, ni, weighted, ax, lds4, POLE_CHUNK);
    if (threadIdx.x == 0) out2[2 * blockIdx.y + blockIdx.x] = v;
}
/* d_out[2 f] = north, d_out[2 f + 1] = south pole value of field f (fields field_stride floats apart) */
extern "C" int ezhip_polevals_batch(float *d_out, const float *d_zin, size_t field_stride, int nfields, int ni, int nj, int weighted, const float *d_ax)
{
    hipLaunchKernelGGL(k_polevals, dim3(2, nfields), dim3(256), 0, g_stream, d_out, d_zin, field_stride, ni, nj, weighted, d_ax);
    return LAUNCH_CHECK("k_polevals");
}
extern "C" int ezhip_polevals(float *d_out2, const float *d_zin, int ni, int nj, int weighted, const float *d_ax)
{
    return ezhip_polevals_batch(d_out2, d_zin, 0, 1, ni, nj, weighted, d_ax);
}

/* float min/max through order-preserving unsigned keys (no NaN handling: the reference has none) */

__global__ __launch_bounds__(256) void k_minmax(unsigned *keys2, const float *z, size_t n)
{
    unsigned kmin = 0xffffffffu, kmax = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned k = f2key(z[i]);
        kmin = min(kmin, k); kmax = max(kmax, k);
    }
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_down((int)kmin, off, 64));
        kmax = max(kmax, (unsigned)__shfl_down((int)kmax, off, 64));
    }
    /* one atomic pair per BLOCK (same-address device atomics cost ~23 ns each: one pair per wave of a 2048-block grid was
     * 190 us of serialised atomics for a 26 M-point field) */
    __shared__ unsigned smin[4], smax[4];
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = kmin; smax[threadIdx.x >> 6] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&keys2[0], min(min(smin[0], smin[1]), min(smin[2], smin[3])));
        atomicMax(&keys2[1], max(max(smax[0], smax[1]), max(smax[2], smax[3])));
    }
}
/* ez_corrval.c:60-87 fill value from min/max */
__global__ void k_fill(float *fill, const unsigned *keys2, int degre_extrap, float valeur, int vector_mode)
{
    float vmin = key2f(keys2[0]), vmax = key2f(keys2[1]);
    float f = 0.0f;
    if (!vector_mode) {
        if (degre_extrap == 4) f = (float)((double)vmax + 0.05 * (double)(vmax - vmin));
        else if (degre_extrap == 5) f = (float)((double)vmin - 0.05 * (double)(vmax - vmin));
        else if (degre_extrap == 6) f = valeur;
    }
    *fill = f;
}
extern "C" int ezhip_fill_value(float *d_fill, const float *d_zin, size_t n, int degre_extrap, float valeur, int vector_mode)
{
    /* d_fill points at float[4]: [0] = fill, [2..3] reused as the two uint keys */
    unsigned *keys = (unsigned *)(d_fill + 2);
    unsigned init[2] = {0xffffffffu, 0u};
    if (set_err(hipMemcpyAsync(keys, init, sizeof(init), hipMemcpyHostToDevice, g_stream), "fill init")) return -1;
    if (!vector_mode && (degre_extrap == 4 || degre_extrap == 5)) {
        int nb = (int)((n + 255) / 256); if (nb > 512) nb = 512;
        hipLaunchKernelGGL(k_minmax, dim3(nb), dim3(256), 0, g_stream, keys, d_zin, n);
    }
    hipLaunchKernelGGL(k_fill, dim3(1), dim3(1), 0, g_stream, d_fill, keys, degre_extrap, valeur, vector_mode);
    return LAUNCH_CHECK("k_fill");
}

/* ===================================================================================== */
/* k_locate                                                                                 */
/* ===================================================================================== */
__device__ __forceinline__ int d_cherche(float val, const float *tab, int n)
{   /* ez_cherche.inc:53-69 */
    int debut = 1, fin = n;
    int milieu = (int)((float)(debut + fin) * 0.5f);
    while (milieu != debut) {
        if (val <= tab[milieu - 1]) fin = milieu; else debut = milieu;
        milieu = (int)((float)(debut + fin) * 0.5f);
    }
    return milieu;
}

/* true (lon,lat) -> rotated (lon,lat): ez_lac.inc + mxm + ez_cal.inc with matrix r */
/* (lon, lat) -> the rotated frame, from the REAL cos / sin of the two angles (ez_lac + mxm + ez_cal) */
__device__ __forceinline__ void d_rotate_cs(const float *r, float coslat, float sinlat, float coslon, float sinlon, float &lon_o, float &lat_o);
__device__ __forceinline__ void d_rotate(const float *r, float lon, float lat, float &lon_o, float &lat_o)
{
    const float dar = (float)(3.14159274101257324 / 180.0);     /* acos(-1.)/180. evaluated in REAL */
    d_rotate_cs(r, glx_cosf(dar * lat), glx_sinf(dar * lat), glx_cosf(dar * lon), glx_sinf(dar * lon), lon_o, lat_o);
}
/* the same with the C library's own REAL functions (libm_exact.h: GNU libc 2.35's sinf / cosf / asinf / atan2f operation by operation): the locate of a
 * rotated source then has the bits ez_gfxyfll has on the host (ez_gfxyfll.c:38-57, ez_lac.inc:31-47, ez_cal.inc:22-47) */
__device__ __noinline__ void d_rotate_exact(const float *r, float lon, float lat, float &lon_o, float &lat_o)
{
    const float dar = (float)(3.14159274101257324 / 180.0);
    const float cosdar = glx_cosf(dar * lat);
    const float c0 = cosdar * glx_cosf(dar * lon), c1 = cosdar * glx_sinf(dar * lon), c2 = glx_sinf(dar * lat);
    float q[3];
    for (int i = 0; i < 3; i++) { float s = 0.0f; s = s + r[i] * c0; s = s + r[3 + i] * c1; s = s + r[6 + i] * c2; q[i] = s; }
    const float rad = (float)(180.0 / 3.14159274101257324);
    lat_o = glx_asinf(fmaxf(-1.00f, fminf(1.0f, q[2]))) * rad;
    float lo = glx_atan2f(q[1], q[0]) * rad;
    lo = fmodf(lo, 360.0f);
    if (lo < 0.0f) lo = lo + 360.0f;
    lon_o = lo;
}
__device__ __forceinline__ void d_rotate_cs(const float *r, float coslat, float sinlat, float coslon, float sinlon, float &lon_o, float &lat_o)
{
    float cosdar = coslat;
    float c0 = cosdar * coslon, c1 = cosdar * sinlon, c2 = sinlat;
    float q[3];
    for (int i = 0; i < 3; i++) { float s = 0.0f; s = s + r[i] * c0; s = s + r[3 + i] * c1; s = s + r[6 + i] * c2; q[i] = s; }
    const float rad = (float)(180.0 / 3.14159274101257324);
    lat_o = glx_asinf(fmaxf(-1.00f, fminf(1.0f, q[2]))) * rad;
    float lo = glx_atan2f(q[1], q[0]) * rad;
    lo = fmodf(lo, 360.0f);
    if (lo < 0.0f) lo = lo + 360.0f;
    lon_o = lo;
}

__global__ __launch_bounds__(256) void k_locate(ezhip_locate_plan p, float *__restrict__ xo, float *__restrict__ yo,
                                                const float *__restrict__ lats, const float *__restrict__ lons,
                                                int ni_dst, int nj_dst, int separable)
{
    size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t npts = (size_t)ni_dst * nj_dst;
    if (n >= npts) return;
    float lat, lon;
    if (separable) { lat = lats[n / ni_dst]; lon = lons[n % ni_dst]; }
    else { lat = lats[n]; lon = lons[n]; }
    float px, py;
    if (p.kind == 4) {                                      /* ez_vxyfll.inc:32-58: REAL dgtord products, the rest in double */
        const float dgtord = 1.7453292519943e-2f;
        const float pi = p.lat0, pj = p.lon0, d60 = p.dlat, dgrw = p.dlon;
        const double re = 1.866025 * 6.371e+6 / (double)d60;
        double rlon, rlat;
        if (p.lon_fix == 1) {
            rlon = (double)(float)(dgtord * (float)(lon + dgrw));
            rlat = (double)(float)(dgtord * lat);
        } else {
            rlon = (double)lon;
            if (rlon > 180.0) rlon = rlon - 360.0;
            rlon = (double)dgtord * (-rlon + (double)dgrw);
            rlat = (double)(float)(dgtord * (-lat));
        }
        const double sinlat = sin(rlat);
        const double r = re * sqrt((1.0 - sinlat) / (1.0 + sinlat));
        px = (float)(r * cos(rlon) + (double)pi);
        py = (float)(r * sin(rlon) + (double)pj);
    } else if (p.kind == 0 || p.kind == 3) {
        if (p.kind == 3) { float lo, la; d_rotate_exact(p.r, lon, lat, lo, la); lon = lo; lat = la; }
        if (p.lon_fix == 1) {                               /* ez_ll2rgd.inc:137-145 */
            if (lon < p.lon0) lon = lon + 360.0f;
            if (lon > (p.lon0 + (float)p.ni * p.dlon)) lon = lon - 360.0f;
        } else if (p.lon_fix == 2) {
            if (lon < 0.0f) lon = lon + 360.0f;
        }
        if (lon < 0.0f) lon = lon + 360.0f;                 /* ez_llll2gd.inc:40-45 (lonref 0) */
        px = (lon - p.lon0) / p.dlon + 1.0f;
        py = (lat - p.lat0) / p.dlat + 1.0f;
    } else {
        if (p.kind == 1) {                                  /* ez_ll2igd.inc:55-66 */
            if (p.lonref == -180.0f) { if (lon > 180.0f) lon = lon - 360.0f; }
            else { if (lon < 0.0f) lon = lon + 360.0f; }
            px = (lon - p.lon0) / p.dlon + 1.0f;
            py = (lat - p.lat0) / p.dlat + 1.0f;
            px = px - 1.0f; py = py - 1.0f;
        } else {                                            /* ez_ll2igd.inc:69-72 */
            d_rotate_exact(p.r, lon, lat, px, py);
        }
        int indx = d_cherche(px, p.ax, p.ni);               /* ez_ll2igd.inc:74-85 */
        int indy = d_cherche(py, p.ay, p.nj);
        if (indx >= p.ni) indx = p.ni - 1;
        if (indy >= p.nj) indy = p.nj - 1;
        px = (float)indx + (px - p.ax[indx - 1]) / (p.ax[indx] - p.ax[indx - 1]);
        py = (float)indy + (py - p.ay[indy - 1]) / (p.ay[indy] - p.ay[indy - 1]);
    }
    xo[n] = px; yo[n] = py;
}

/* libm_exact.h on the device, function by function (fn 0 sinf, 1 cosf, 2 asinf, 3 atanf, 4 atan2f(a, b)): what tests/test_gpu_interp.py and
 * tools/check_libm_exact_gpu.py compare with the host's C library */
__global__ __launch_bounds__(256) void k_libm_exact_probe(int fn, const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, size_t n)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const float x = a[k];
    out[k] = fn == 0 ? glx_sinf(x) : fn == 1 ? glx_cosf(x) : fn == 2 ? glx_asinf(x) : fn == 3 ? glx_atanf(x) : glx_atan2f(x, b[k]);
}
extern "C" int ezhip_libm_exact_probe(int fn, const float *d_a, const float *d_b, float *d_out, size_t n)
{
    if (!n) return 0;
    if (fn < 0 || fn > 4 || (fn == 4 && !d_b)) return -1;
    hipLaunchKernelGGL(k_libm_exact_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, fn, d_a, d_b, d_out, n);
    return LAUNCH_CHECK("k_libm_exact_probe");
}

/* does any located point fall outside the source (the DEHORS zone's test, ez_defzone_dehors.c:63-74: nint of x, y against 1 .. ni, 1 .. nj)? *flag |= 1 */
__global__ __launch_bounds__(256) void k_any_dehors(const float *__restrict__ x, const float *__restrict__ y, size_t n, int ni, int nj, int *flag)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    int out = 0;
    if (k < n) {
        const int ix = (int)((double)x[k] + 0.5), iy = (int)((double)y[k] + 0.5);
        out = ix < 1 || iy < 1 || ix > ni || iy > nj;
    }
    if (__ballot(out) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
extern "C" int ezhip_any_dehors(const float *d_x, const float *d_y, size_t n, int ni, int nj, int *d_flag)
{
    if (hipMemsetAsync(d_flag, 0, sizeof(int), g_stream) != hipSuccess) return -1;
    if (!n) return 0;
    hipLaunchKernelGGL(k_any_dehors, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, d_x, d_y, n, ni, nj, d_flag);
    return LAUNCH_CHECK("k_any_dehors");
}

extern "C" int ezhip_locate(const ezhip_locate_plan *plan, float *d_x, float *d_y,
                            const float *d_lat, const float *d_lon, int ni_dst, int nj_dst, int separable)
{
    size_t npts = (size_t)ni_dst * nj_dst;
    if (!npts) return 0;
    hipLaunchKernelGGL(k_locate, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, g_stream, *plan, d_x, d_y, d_lat, d_lon, ni_dst, nj_dst, separable);
    return LAUNCH_CHECK("k_locate");
}

/* ===================================================================================== */
/* k_wind_rotate                                                                            */
/* ===================================================================================== */
/* {cos, sin}(dar * angle) in fp64 for the n angles of a separable target's longitudes (or latitudes): the same expressions
 * k_wind_rotate evaluates per point, once per column / row */
__global__ __launch_bounds__(256) void k_wind_trig(double *tab, float *tabf, const float *ang, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double dar = (double)(float)(3.14159274101257324 / 180.0);
    tab[2 * i] = cos(dar * (double)ang[i]); tab[2 * i + 1] = sin(dar * (double)ang[i]);
    const float darf = (float)(3.14159274101257324 / 180.0);        /* the REAL pair d_rotate uses */
    tabf[2 * i] = glx_cosf(darf * ang[i]); tabf[2 * i + 1] = glx_sinf(darf * ang[i]);
}
extern "C" int ezhip_wind_trig_tables(double *d_lon_trig, double *d_lat_trig, float *d_lon_trigf, float *d_lat_trigf,
                                      const float *d_lat, const float *d_lon, int ni, int nj)
{
    hipLaunchKernelGGL(k_wind_trig, dim3((ni + 255) / 256), dim3(256), 0, g_stream, d_lon_trig, d_lon_trigf, d_lon, ni);
    hipLaunchKernelGGL(k_wind_trig, dim3((nj + 255) / 256), dim3(256), 0, g_stream, d_lat_trig, d_lat_trigf, d_lat, nj);
    return LAUNCH_CHECK("k_wind_trig");
}

/* c_ezgfwfllw after its ez_gdwfllw 'L' step: true components (u, v) at true (lon, lat) -> components on the rotated target grid:
 * ez_uvacart at the TRUE coordinates, mxm with r, ez_cartauv at the ROTATED coordinates (ez_gfxyfll of the true ones) */
__device__ __forceinline__ void d_to_rotated_target(const float *r, float lon, float lat, float &u, float &v)
{
    const double dar = (double)(float)(3.14159274101257324 / 180.0);
    double a, b, c, d;
    sincos(dar * (double)lon, &a, &b);
    sincos(dar * (double)lat, &c, &d);
    const float x0 = (float)(-((double)u * a) - ((double)v * b * c));
    const float x1 = (float)(((double)u * b) - ((double)v * a * c));
    const float x2 = (float)((double)v * d);
    float q[3];
    for (int i = 0; i < 3; i++) { float s = 0.0f; s = s + r[i] * x0; s = s + r[3 + i] * x1; s = s + r[6 + i] * x2; q[i] = s; }
    float lon_r, lat_r;
    d_rotate(r, lon, lat, lon_r, lat_r);
    const double aa = cos(dar * (double)lon_r), bb = sin(dar * (double)lon_r), ee = cos(dar * (double)lat_r), ff = sin(dar * (double)lat_r);
    u = (float)(((double)q[1] * aa) - ((double)q[0] * bb));
    const double cc = ((double)q[0] * aa) + ((double)q[1] * bb);
    const double dd = sqrt(cc * cc + (double)(q[2] * q[2]));
    const double sg = ((double)q[2] * ee) - (cc * ff);
    v = (float)(sg >= 0.0 ? fabs(dd) : -fabs(dd));
}

__global__ __launch_bounds__(256) void k_wind_rotate(ezhip_wind_plan p, float *__restrict__ uu, float *__restrict__ vv,
                                                     const float *__restrict__ lats, const float *__restrict__ lons,
                                                     int ni_dst, int nj_dst)
{
    const float RDTODG = 57.295779513082f, DGTORD = 1.7453292519943e-2f;
    size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t npts = (size_t)ni_dst * nj_dst;
    if (n >= npts) return;
    float lat, lon;
    if (p.separable) { lat = lats[n / ni_dst]; lon = lons[n % ni_dst]; }
    else { lat = lats[n]; lon = lons[n]; }
    float u = uu[n], v = vv[n];
    const size_t li = p.separable ? n % (size_t)ni_dst : n;         /* index of the point's longitude */
    if (p.wd_in) {                                          /* speed / direction given: c_gduvfwd only */
        const float spd_ = u, dir_ = v;
        float psi_ = p.dst_ps == 1 ? lon + p.dst_xg4 - dir_ : p.dst_ps == 2 ? 180.0f - lon + p.dst_xg4 - dir_ : 270.0f - dir_;
        float uo = glx_cosf(psi_ * DGTORD) * spd_, vo = glx_sinf(psi_ * DGTORD) * spd_;
        if (p.dst_rotated) d_to_rotated_target(p.r_dst, lon, lat, uo, vo);
        if (p.dst_lamb_cs) { const float c_ = p.dst_lamb_cs[2 * li], s_ = p.dst_lamb_cs[2 * li + 1], a_ = uo, b_ = vo; uo = a_ * c_ - b_ * s_; vo = a_ * s_ + b_ * c_; }      /* ez_lamb_gdwfllw.inc:52-56 */
        uu[n] = uo; vv[n] = vo;
        return;
    }
    if (p.src_rotated) {                                    /* c_ezllwfgfw, ez_llwfgfw.c:38-73 */
        float lon_r, lat_r;
        if (p.separable && p.lon_trigf) {       /* REAL cos / sin of the column's longitude and the row's latitude from tables */
            const size_t jr = n / ni_dst, ic = n - jr * ni_dst;
            d_rotate_cs(p.r, p.lat_trigf[2 * jr], p.lat_trigf[2 * jr + 1], p.lon_trigf[2 * ic], p.lon_trigf[2 * ic + 1], lon_r, lat_r);
        } else d_rotate(p.r, lon, lat, lon_r, lat_r);
        const double dar = (double)(float)(3.14159274101257324 / 180.0);
        double a, b, c, d;
        if (p.fast_trig) {      /* REAL sine / cosine of the rotated coordinates (1e-7 relative, the tolerance is 1e-5): 4x fewer fp64 operations */
            float af, bf, cf, df;
            sincosf((float)dar * lon_r, &af, &bf); sincosf((float)dar * lat_r, &cf, &df);
            a = af; b = bf; c = cf; d = df;
        } else {
            sincos(dar * (double)lon_r, &a, &b);
            sincos(dar * (double)lat_r, &c, &d);
        }
        float x0 = (float)(-((double)u * a) - ((double)v * b * c));     /* ez_uvacart.inc */
        float x1 = (float)(((double)u * b) - ((double)v * a * c));
        float x2 = (float)((double)v * d);
        float q[3];
        for (int i = 0; i < 3; i++) { float s = 0.0f; s = s + p.ri[i] * x0; s = s + p.ri[3 + i] * x1; s = s + p.ri[6 + i] * x2; q[i] = s; }
        double aa, bb, ee, ff;                                             /* ez_cartauv.inc at TRUE lon/lat */
        if (p.separable && p.lon_trig) {     /* separable target: cos/sin of a column's longitude / a row's latitude from tables */
            const size_t jr = n / ni_dst, ic = n - jr * ni_dst;
            aa = p.lon_trig[2 * ic]; bb = p.lon_trig[2 * ic + 1]; ee = p.lat_trig[2 * jr]; ff = p.lat_trig[2 * jr + 1];
        } else {
            aa = cos(dar * (double)lon); bb = sin(dar * (double)lon);
            ee = cos(dar * (double)lat); ff = sin(dar * (double)lat);
        }
        u = (float)(((double)q[1] * aa) - ((double)q[0] * bb));
        double cc = ((double)q[0] * aa) + ((double)q[1] * bb);
        double dd = sqrt(cc * cc + (double)(q[2] * q[2]));
        double sg = ((double)q[2] * ee) - (cc * ff);
        v = (float)(sg >= 0.0 ? fabs(dd) : -fabs(dd));
    }
    if (p.src_lamb_cs) {                                    /* ez_lamb_llwfgdw.inc:51-56: grid components -> true east / north components, then as 'L' (:58-75) */
        const float c_ = p.src_lamb_cs[2 * li], s_ = p.src_lamb_cs[2 * li + 1], a_ = u, b_ = v;
        u = a_ * c_ - b_ * s_; v = a_ * s_ + b_ * c_;
    }
    /* components -> speed, direction: ez_llwfgdw.inc:91-114 ('N'), :117-140 ('S'), :143-165 ('L'/A/B/G) */
    float spd = sqrtf(u * u + v * v), dir;
    if (spd == 0.0f) dir = 0.0f;
    else if (p.src_ps == 1) dir = (u == 0.0f) ? ((v >= 0.0f) ? lon + p.src_xg4 - 90.0f : lon + p.src_xg4 + 90.0f) : lon + p.src_xg4 - RDTODG * glx_atan2f(v, u);
    else if (p.src_ps == 2) dir = (u == 0.0f) ? ((v >= 0.0f) ? 90.0f - lon + p.src_xg4 : 270.0f - lon + p.src_xg4) : 180.0f - lon + p.src_xg4 - RDTODG * glx_atan2f(v, u);
    else if (u == 0.0f) dir = (v >= 0.0f) ? 180.0f : 0.0f;
    else dir = 270.0f - RDTODG * glx_atan2f(v, u);
    dir = d_fmod360(d_fmod360(dir) + 360.0f);
    if (p.wd_only) { uu[n] = spd; vv[n] = dir; return; }      /* c_ezwdint: speed / direction are the result */
    /* speed, direction -> target components: ez_gdwfllw.inc:93-105 ('N'), :108-121 ('S'), :123-134 ('L'/A/B/G) */
    float psi = p.dst_ps == 1 ? lon + p.dst_xg4 - dir : p.dst_ps == 2 ? 180.0f - lon + p.dst_xg4 - dir : 270.0f - dir;
    float uo = glx_cosf(psi * DGTORD) * spd, vo = glx_sinf(psi * DGTORD) * spd;
    if (p.dst_rotated) d_to_rotated_target(p.r_dst, lon, lat, uo, vo);      /* Z-on-E target: c_ezgfwfllw */
    if (p.dst_lamb_cs) { const float c_ = p.dst_lamb_cs[2 * li], s_ = p.dst_lamb_cs[2 * li + 1], a_ = uo, b_ = vo; uo = a_ * c_ - b_ * s_; vo = a_ * s_ + b_ * c_; }          /* '!' target: ez_lamb_gdwfllw.inc:52-56 */
    uu[n] = uo; vv[n] = vo;
}

/* ===================================================================================== */
/* masks: qqq_ezsint_mask / qqq_ezget_mask_zones (ezget_mask_zones.inc), lorenzo_mask_fill method 2  */
/* ===================================================================================== */
/* mask_in(i1, j1), 1-based, with the Fortran's linear addressing (no bounds in the reference: an address past the
 * array is clamped to it here) */
__device__ __forceinline__ int mask_at(const int *m, int ni, int nj, int i1, int j1)
{
    long long k = (long long)(j1 - 1) * ni + (i1 - 1), n = (long long)ni * nj;
    k = k < 0 ? 0 : (k >= n ? n - 1 : k);
    return m[k];
}
/* mode 0: c_ezsint_mask (cloud_linear: the second pass of ezget_mask_zones.inc:91-103, which only reads the point's own
 * first-pass value); mode 1: c_ezget_mask_zones */
__global__ __launch_bounds__(256) void k_mask(int *__restrict__ mask_out, const float *__restrict__ x, const float *__restrict__ y,
                                              const int *__restrict__ mask_in, int ni_in, int nj_in, int ni_out, int nj_out, int mode, int cloud_linear)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= (size_t)ni_out * nj_out) return;
    const float xv = x[k], yv = y[k];
    const int ix = (int)xv, iy = (int)yv;
    const bool outside = ix < 1 || ix > ni_in || iy < 1 || iy > nj_in;
    if (mode == 1) {
        if (outside) { mask_out[k] = 7; return; }
        int nmissing = 0;
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) nmissing += mask_at(mask_in, ni_in, nj_in, ix + a, iy + b) == 0;
        mask_out[k] = 4 - nmissing;
        return;
    }
    int v = 1;
    if (outside) v = 0;
    else if (mask_at(mask_in, ni_in, nj_in, (int)lroundf(xv), (int)lroundf(yv)) == 0) v = 0;
    if (cloud_linear && v == 1) {
        const int j = (int)(k / ni_out), i = (int)(k - (size_t)j * ni_out);
        if (i < ni_out - 1 && j < nj_out - 1 &&
            (mask_at(mask_in, ni_in, nj_in, ix + 1, iy) == 0 || mask_at(mask_in, ni_in, nj_in, ix, iy + 1) == 0 || mask_at(mask_in, ni_in, nj_in, ix + 1, iy + 1) == 0)) v = 0;
    }
    mask_out[k] = v;
}
extern "C" int ezhip_mask(int *d_mask_out, const float *d_x, const float *d_y, const int *d_mask_in, int ni_in, int nj_in, int ni_out, int nj_out, int mode, int cloud_linear)
{
    size_t n = (size_t)ni_out * nj_out;
    if (!n) return 0;
    hipLaunchKernelGGL(k_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, d_mask_out, d_x, d_y, d_mask_in, ni_in, nj_in, ni_out, nj_out, mode, cloud_linear);
    return LAUNCH_CHECK("k_mask");
}
__global__ __launch_bounds__(256) void k_mask_fill(float *__restrict__ fld, const int *__restrict__ mask, const unsigned *keys2, size_t n)
{
    const float rmin = key2f(keys2[0]);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) if (mask[i] == 0) fld[i] = rmin;
}
/* lorenzo_mask_fill(fld, mask, ni, nj, 2): masked points take minval(fld).  d_keys2: two words of device scratch */
extern "C" int ezhip_mask_fill_min(float *d_fld, const int *d_mask, size_t n, unsigned *d_keys2)
{
    if (!n) return 0;
    unsigned init[2] = {0xffffffffu, 0u};
    if (set_err(hipMemcpyAsync(d_keys2, init, sizeof(init), hipMemcpyHostToDevice, g_stream), "mask fill init")) return -1;
    int nb = (int)((n + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_minmax, dim3(nb > 512 ? 512 : nb), dim3(256), 0, g_stream, d_keys2, d_fld, n);
    hipLaunchKernelGGL(k_mask_fill, dim3(nb), dim3(256), 0, g_stream, d_fld, d_mask, d_keys2, n);
    return LAUNCH_CHECK("k_mask_fill");
}

/* ez_xpngdag2.inc / ez_xpngdb2.inc: row jo (j1..j2) of the expansion <- source row (1..nj), mirrored rows times sign */
__global__ __launch_bounds__(256) void k_hemi_expand(float *__restrict__ dst, const float *__restrict__ src, int ni, int nj, int j1, int j2, int hem, int is_b, float sign, int yinv)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)ni * (size_t)(j2 - j1 + 1);
    if (k >= n) return;
    const int r = (int)(k / ni), i = (int)(k - (size_t)r * ni), jo = j1 + r;
    int js; float sg = 1.0f;
    if (jo >= 1 && jo <= nj) js = jo;
    else if (hem == 1) { js = is_b ? 2 - jo : 1 - jo; sg = sign; }          /* NORD: zout(i, 2-j) / zout(i, -j+1) = sign * zi(i, j) */
    else { js = is_b ? 2 * nj - jo : 2 * nj + 1 - jo; sg = sign; }          /* SUD:  zout(i, nj+j-1) / zout(i, nj+j) = sign * zi(i, nj-j+1) */
    if (yinv) js = nj + 1 - js;                               /* the source rows were reversed first (PERMUT) */
    dst[k] = sg * src[(size_t)(js - 1) * ni + i];
}
extern "C" int ezhip_hemi_expand(float *d_dst, const float *d_src, int ni, int nj, int j1, int j2, int hem, int is_b, int symetrie, int yinv)
{
    const size_t n = (size_t)ni * (size_t)(j2 - j1 + 1);
    hipLaunchKernelGGL(k_hemi_expand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, d_dst, d_src, ni, nj, j1, j2, hem, is_b, symetrie == 0 ? -1.0f : 1.0f, yinv);
    return LAUNCH_CHECK("k_hemi_expand");
}

__global__ __launch_bounds__(256) void k_scatter(float *__restrict__ dst, const float *__restrict__ src, const int *__restrict__ idx, int n)
{
    int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) dst[idx[k]] = src[k];
}
extern "C" int ezhip_scatter(float *d_dst, const float *d_src, const int *d_idx, int n)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, d_dst, d_src, d_idx, n);
    return LAUNCH_CHECK("k_scatter");
}

/* interp_degree = "average" (ez_avg.inc:81-211): one thread per target cell; the cell's bounds in source index space come from the host
 * (the reference derives them from the first target row / column only); the source cells under it are added in row-major order with the
 * covered fraction as weight, all in REAL, multiply and add apart.  The three row regimes (first / middle / last target row) differ in how
 * the first and last source rows are cut.  Columns are addressed modulo the source's wrap (ztmp, :19-52); rows and, on a regional source,
 * columns beyond the field -- where the reference indexes outside its array -- are clamped. */
__device__ __forceinline__ float avg_src(const float *__restrict__ zin, int ni, int nj, int ext, int ii, int jj)
{
    jj = min(max(jj, 1), nj);
    if (ext == 1) { int k = ii; while (k < 1) k += ni - 1; while (k > ni - 1) k -= ni - 1; ii = k; }
    else if (ext == 2) { int k = ii; while (k < 1) k += ni; while (k > ni) k -= ni; ii = k; }
    else ii = min(max(ii, 1), ni);
    return zin[(size_t)(jj - 1) * ni + (ii - 1)];
}
__global__ __launch_bounds__(256) void k_average(float *__restrict__ zout, const float *__restrict__ zin, const float *__restrict__ bounds,
                                                 int nid, int njd, int nis, int njs, int ext, float ylast)
{
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= (size_t)nid * njd) return;
    const int j = (int)(n / nid), i = (int)(n - (size_t)j * nid);
    /* bounds = [x nid | widening njd | y_low njd | y_high njd]: the cell reaches half way to its neighbours in x, times the row's widening
     * (1 for "average"; 1 / cos(latitude), evaluated on the host, for "sph_average": ez_avg_sph.inc:63-93) */
    const float *xs = bounds, amp = bounds[nid + j], yl = bounds[nid + njd + j], yh = bounds[nid + 2 * njd + j];
    const float xl = i == 0 ? xs[0] - 0.5f * (xs[1] - xs[0]) * amp : xs[i] - 0.5f * (xs[i] - xs[i - 1]) * amp;
    const float xh = i == nid - 1 ? xs[nid - 1] + 0.5f * (xs[nid - 1] - xs[nid - 2]) * amp : xs[i] + 0.5f * (xs[i + 1] - xs[i]) * amp;
    const int row = j == 0 ? 0 : (j == njd - 1 ? 2 : 1);
    int jstart, jend = (int)lroundf(yh), istart = (int)xl, iend = (int)lroundf(xh);
    if (row == 2) jstart = (int)ylast;
    else {
        jstart = (int)yl;
        if (row == 1) { if ((0.5f + (float)jstart) < yl) jstart = jstart + 1; }
        else { if ((float)jstart > yl) jstart = jstart - 1; }
    }
    if ((0.5f + (float)istart) < xl) istart = istart + 1;
    if (row == 0 && (float)iend < xh) iend = iend + 1;
    float z = 0.0f, total = 0.0f;
    for (int jj = jstart; jj <= jend; jj++) {
        float ymin = (float)jj - 0.5f, ymax = (float)jj + 0.5f, yfrac = 1.0f;
        if (row == 0) { if (jj == 1) ymin = 1.0f; yfrac = ymax - ymin; }
        if (row == 2) { if (jj == njs) ymax = (float)njs; yfrac = ymax - ymin; }
        if (ymin < yl) yfrac = ymax - yl;
        if (ymax > yh) yfrac = yh - ymin;
        for (int ii = istart; ii <= iend; ii++) {
            const float xmin = (float)ii - 0.5f, xmax = (float)ii + 0.5f;
            float xfrac = 1.0f;
            if (xmin < xl) xfrac = xmax - xl;
            if (xmax > xh) xfrac = xh - xmin;
            const float area = xfrac * yfrac;
            total = total + area;
            const float prod = avg_src(zin, nis, njs, ext, ii, jj) * area;
            z = z + prod;
        }
    }
    if (total != 0.0f) z = z / total;
    zout[n] = z;
}
extern "C" int ezhip_average(float *d_zout, const float *d_zin, const float *d_bounds, int nid, int njd, int nis, int njs, int ext, float ylast)
{
    const size_t n = (size_t)nid * njd;
    if (!n) return 0;
    hipLaunchKernelGGL(k_average, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_stream, d_zout, d_zin, d_bounds, nid, njd, nis, njs, ext, ylast);
    return LAUNCH_CHECK("k_average");
}

/* The whole wind chain of a grid pair (grid-frame components -> true components -> speed / direction -> target components) is, point by
 * point, a plane rotation that depends on the two grids only: ezhip_wind_matrix runs the chain once on the unit vectors (1,0) and (0,1)
 * and keeps the four coefficients per point; k_wind_apply then replaces ~10 REAL*8 / REAL transcendentals per point and call by two
 * multiply-adds per component.  The result differs from the chain on (u,v) by the chain's own rounding (~4e-7 |V|; tolerance 1e-5 |V|). */
/* M: the four coefficients per point; M2 (behind it): the rotation (a + d) / 2, (b - c) / 2 the four stand for when the chain is one, as one word per point (rot_pack); *dev: the largest
 * distance of a point's matrix from that form (max |a - d|, |b + c|; infinite when a coefficient is not finite), as the bits of a non-negative float */
__global__ __launch_bounds__(256) void k_wind_matrix_pack(float4 *__restrict__ M, unsigned *__restrict__ M2, unsigned *__restrict__ dev, const float *__restrict__ a,
                                                          const float *__restrict__ c, const float *__restrict__ b, const float *__restrict__ d, size_t n)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    float dv = 0.0f;
    if (k < n) {
        const float ak = a[k], bk = b[k], ck = c[k], dk = d[k];
        M[k] = make_float4(ak, bk, ck, dk);                          /* uo = a u + b v, vo = c u + d v */
        M2[k] = rot_pack(0.5f * (ak + dk), 0.5f * (bk - ck));
        dv = fmaxf(fabsf(ak - dk), fabsf(bk + ck));
        if (!(dv <= 3.0e38f)) dv = __builtin_inff();
    }
    for (int off = 32; off; off >>= 1) dv = fmaxf(dv, __shfl_xor(dv, off));
    /* one address for 125 000 waves: the atomics went through the L2 one after the other (1.4 ms for cfg3's 8 M points).  The maximum only grows: a wave whose
     * value does not exceed what is there already has nothing to add -- a plain (cached) read tells it */
    if ((threadIdx.x & 63) == 0 && dv > 0.0f && __float_as_uint(dv) > __atomic_load_n(dev, __ATOMIC_RELAXED)) atomicMax(dev, __float_as_uint(dv));
}
__global__ __launch_bounds__(256) void k_fill2(float *__restrict__ a, float va, float *__restrict__ b, float vb, size_t n)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) { a[k] = va; b[k] = vb; }
}
__global__ __launch_bounds__(256) void k_wind_apply(const void *__restrict__ M, int half, float *__restrict__ uu, float *__restrict__ vv, size_t n, int dst_rot)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    wm_f2 lo, hi;
    wind_m_load(M, half, k, lo, hi);
    const float u = uu[k], v = vv[k];
    float a, b;
    wind_m_apply(lo, hi, half, u, v, dst_rot, a, b);
    uu[k] = a; vv[k] = b;
}
/* d_M: 20 bytes per point (the float4 form, then the packed rotations); *max_dev: how far the matrices are from pure rotations (see k_wind_matrix_pack) */
extern "C" int ezhip_wind_matrix(const ezhip_wind_plan *plan, void *d_M, const float *d_lat, const float *d_lon, int ni_dst, int nj_dst, float *max_dev)
{
    const size_t npts = (size_t)ni_dst * nj_dst;
    *max_dev = 0.0f;
    if (!npts) return 0;
    float *t = nullptr;
    if (set_err(hipMalloc((void **)&t, sizeof(float) * 4 * npts + 16), "wind matrix scratch")) return -1;
    unsigned *d_dev = (unsigned *)(t + 4 * npts);
    if (hipMemsetAsync(d_dev, 0, 4, g_stream) != hipSuccess) { (void)hipFree(t); return -1; }
    const dim3 grid((unsigned)((npts + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_fill2, grid, block, 0, g_stream, t, 1.0f, t + npts, 0.0f, npts);                      /* (1,0) -> (a, c) */
    hipLaunchKernelGGL(k_fill2, grid, block, 0, g_stream, t + 2 * npts, 0.0f, t + 3 * npts, 1.0f, npts);       /* (0,1) -> (b, d) */
    hipLaunchKernelGGL(k_wind_rotate, grid, block, 0, g_stream, *plan, t, t + npts, d_lat, d_lon, ni_dst, nj_dst);
    hipLaunchKernelGGL(k_wind_rotate, grid, block, 0, g_stream, *plan, t + 2 * npts, t + 3 * npts, d_lat, d_lon, ni_dst, nj_dst);
    hipLaunchKernelGGL(k_wind_matrix_pack, grid, block, 0, g_stream, (float4 *)d_M, (unsigned *)((float4 *)d_M + npts), d_dev, t, t + npts, t + 2 * npts, t + 3 * npts, npts);
    int rc = LAUNCH_CHECK("k_wind_matrix");
    unsigned dev_bits = 0x7F800000u;
    if (hipMemcpyAsync(&dev_bits, d_dev, 4, hipMemcpyDeviceToHost, g_stream) != hipSuccess || hipStreamSynchronize(g_stream) != hipSuccess) rc = -1;
    memcpy(max_dev, &dev_bits, 4);
    (void)hipFree(t);
    return rc;
}
extern "C" int ezhip_wind_apply(const void *d_M, int half, float *d_uu, float *d_vv, size_t npts, int dst_rotated)
{
    if (!npts) return 0;
    hipLaunchKernelGGL(k_wind_apply, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, g_stream, d_M, half, d_uu, d_vv, npts, dst_rotated);
    return LAUNCH_CHECK("k_wind_apply");
}

extern "C" int ezhip_wind_rotate(const ezhip_wind_plan *plan, float *d_uu, float *d_vv,
                                 const float *d_lat, const float *d_lon, int ni_dst, int nj_dst)
{
    size_t npts = (size_t)ni_dst * nj_dst;
    if (!npts) return 0;
    hipLaunchKernelGGL(k_wind_rotate, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, g_stream, *plan, d_uu, d_vv, d_lat, d_lon, ni_dst, nj_dst);
    return LAUNCH_CHECK("k_wind_rotate");
}
