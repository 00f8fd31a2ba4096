// TEST INFRASTRUCTURE: scheduler of the "HIP on the CPU" shim (see hip/hip_runtime.h).
// A workgroup runs on ONE OS thread as a set of cooperative fibers (hand-written x86-64 switch), one per GPU thread: __syncthreads() and the
// wave64 collectives are yields, so a barrier costs a context switch instead of a kernel futex and a kernel without barriers
// runs its threads back to back.  Workgroups are distributed over a few OS worker threads (atomics are std::atomic_ref, so
// concurrent workgroups are safe); `__shared__` variables are thread_local statics, i.e. one copy per worker = per workgroup.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdlib>
#include <mutex>

thread_local hipcpu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;
alignas(16) thread_local unsigned char hipcpu_dyn[160 * 1024];
alignas(64) thread_local unsigned char hipcpu_wave_scratch[16][2][64 * 64];
thread_local unsigned char hipcpu_wave_flip[1024];

#if !defined(__x86_64__)
#error "tests/hipcpu/runtime.cpp: the fiber switch below is written for x86-64 (System V ABI)"
#endif
// hipcpu_switch(&save_sp, new_sp): park the callee-saved registers of the running fiber on its stack, continue on `new_sp`.
// (ucontext's swapcontext does the same plus a sigprocmask system call per switch -- millions of them per test.)
extern "C" void hipcpu_switch(void** save_sp, void* new_sp);
asm(R"(
    .text
    .globl hipcpu_switch
    .type hipcpu_switch,@function
hipcpu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipcpu_switch, .-hipcpu_switch
)");

namespace {
constexpr size_t kStack = 512 * 1024, kMaxThreads = 1024;
std::mutex g_pool_mu;
std::vector<char*> g_pool;               // fiber-stack regions (kMaxThreads x kStack, lazily committed), reused across launches

char* pool_get() {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool.empty()) { char* p = g_pool.back(); g_pool.pop_back(); return p; }
    }
    void* p = mmap(nullptr, kMaxThreads * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("hipcpu: mmap"); abort(); }
    return (char*)p;
}
void pool_put(char* p) { std::lock_guard<std::mutex> lk(g_pool_mu); g_pool.push_back(p); }

struct Worker {
    const std::function<void()>* body = nullptr;
    unsigned nt = 0, cur = 0, alive = 0, arrived = 0, gen = 0;
    unsigned wave_alive[16] = {}, wave_arrived[16] = {}, wave_gen[16] = {};
    std::vector<void*> sp;               // parked stack pointer of every fiber
    std::vector<unsigned char> done;
    std::vector<unsigned> wait_on, wait_gen;      // blocked fibers: 0 = runnable, 1 = workgroup barrier, 2 + w = wave w's barrier
    std::vector<hipcpu_idx> tid;
    void* main_sp = nullptr;
    char* stacks = nullptr;
    unsigned long progress = 0;          // bumped whenever a barrier releases or a fiber finishes (deadlock detection)
};
thread_local Worker* g_w = nullptr;

void release_checks(Worker& w, unsigned wave) {
    if (w.alive && w.arrived == w.alive) { w.arrived = 0; ++w.gen; ++w.progress; }
    if (wave < 16 && w.wave_alive[wave] && w.wave_arrived[wave] == w.wave_alive[wave]) { w.wave_arrived[wave] = 0; ++w.wave_gen[wave]; ++w.progress; }
}

void fiber_entry() {
    Worker& w = *g_w;
    (*w.body)();
    const unsigned me = w.cur, wave = me >> 6;
    w.done[me] = 1; --w.alive; ++w.progress;
    if (wave < 16) --w.wave_alive[wave];
    release_checks(w, wave);             // a thread that returned no longer takes part in later barriers
    hipcpu_switch(&w.sp[me], w.main_sp);
    abort();                             // a finished fiber is never resumed
}

void yield_now() {
    Worker& w = *g_w;
    hipcpu_switch(&w.sp[w.cur], w.main_sp);
}

void run_block(Worker& w, dim3 block, unsigned bx, unsigned by, unsigned bz, size_t smem) {
    const unsigned nt = w.nt;
    w.alive = nt; w.arrived = 0;
    for (unsigned v = 0; v < 16; ++v) { w.wave_alive[v] = nt > 64 * v ? std::min(64u, nt - 64 * v) : 0; w.wave_arrived[v] = 0; }
    if (smem) memset(hipcpu_dyn, 0, smem);
    blockIdx = {bx, by, bz};
    for (unsigned t = 0; t < nt; ++t) {
        w.done[t] = 0; w.wait_on[t] = 0; hipcpu_wave_flip[t] = 0;
        void** top = reinterpret_cast<void**>(w.stacks + (size_t)(t + 1) * kStack);      // 16-byte aligned
        *--top = nullptr;                                    // fake return address of fiber_entry (keeps rsp = 8 mod 16 at entry)
        *--top = reinterpret_cast<void*>(&fiber_entry);      // hipcpu_switch's `ret` lands here
        for (int r = 0; r < 6; ++r) *--top = nullptr;        // rbp, rbx, r12..r15
        w.sp[t] = top;
    }
    unsigned long last = w.progress;
    unsigned idle_rounds = 0;
    while (w.alive) {
        for (unsigned t = 0; t < nt; ++t) {
            if (w.done[t]) continue;
            if (const unsigned k = w.wait_on[t]) {                   // still parked at a barrier nobody released: do not even switch
                if ((k == 1 ? w.gen : w.wave_gen[k - 2]) == w.wait_gen[t]) continue;
                w.wait_on[t] = 0;
            }
            w.cur = t; threadIdx = w.tid[t];
            hipcpu_switch(&w.main_sp, w.sp[t]);
        }
        if (w.progress == last) {
            if (++idle_rounds > 4) { fprintf(stderr, "hipcpu: deadlock (a barrier or wave collective not reached by every live thread)\n"); abort(); }
        } else { idle_rounds = 0; last = w.progress; }
    }
}
}  // namespace

unsigned hipcpu_linear_tid() { return g_w->cur; }

void hipcpu_syncthreads() {
    Worker& w = *g_w;
    const unsigned my = w.gen;
    ++w.arrived;
    release_checks(w, 99);
    if (w.gen == my) { w.wait_on[w.cur] = 1; w.wait_gen[w.cur] = my; yield_now(); }
}

void hipcpu_wave_sync() {
    Worker& w = *g_w;
    const unsigned wave = w.cur >> 6;
    if (wave >= 16) { fprintf(stderr, "hipcpu: wave collectives need <= 1024 threads per workgroup\n"); abort(); }
    const unsigned my = w.wave_gen[wave];
    ++w.wave_arrived[wave];
    release_checks(w, wave);
    if (w.wave_gen[wave] == my) { w.wait_on[w.cur] = 2 + wave; w.wait_gen[w.cur] = my; yield_now(); }
}

void hipcpu_run_grid(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    static std::mutex launch_mu;                     // launches are serialised (gridDim / blockDim are process globals)
    std::lock_guard<std::mutex> lk(launch_mu);
    gridDim = grid; blockDim = block;
    const unsigned nt = block.x * block.y * block.z;
    const unsigned long nblocks = (unsigned long)grid.x * grid.y * grid.z;
    if (!nt || !nblocks) return;
    if (nt > kMaxThreads) { fprintf(stderr, "hipcpu: workgroup of %u threads\n", nt); abort(); }
    static const unsigned ncpu = [] { const char* e = getenv("HIPCPU_THREADS"); unsigned n = e ? atoi(e) : std::thread::hardware_concurrency(); return n ? n : 1u; }();
    const unsigned nworkers = (unsigned)std::min<unsigned long>(ncpu, nblocks);
    std::atomic<unsigned long> next{0};
    auto work = [&]() {
        Worker w;
        w.body = &body; w.nt = nt;
        w.sp.resize(nt); w.done.resize(nt); w.tid.resize(nt); w.wait_on.assign(nt, 0); w.wait_gen.assign(nt, 0);
        for (unsigned t = 0; t < nt; ++t) w.tid[t] = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        w.stacks = pool_get();
        g_w = &w;
        for (unsigned long b = next.fetch_add(1); b < nblocks; b = next.fetch_add(1))
            run_block(w, block, (unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long)grid.x * grid.y)), smem);
        g_w = nullptr;
        pool_put(w.stacks);
    };
    static const bool trace = getenv("HIPCPU_TRACE") != nullptr;     // per-launch wall time on stderr
    const auto t0 = std::chrono::steady_clock::now();
    if (nworkers == 1) { std::thread t(work); t.join(); }            // always a fresh thread: fibers must not run on a tiny caller stack
    else {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < nworkers; ++i) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    if (trace)
        fprintf(stderr, "hipcpu: grid %lu x block %u: %.1f ms\n", nblocks, nt, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}
