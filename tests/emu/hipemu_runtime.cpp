// TEST INFRASTRUCTURE: the fiber scheduler of tests/emu/shim/hip/hip_runtime.h (one definition per emulated library).
#include <hip/hip_runtime.h>

#include <thread>
#include <vector>

thread_local hipemu::Idx threadIdx, blockIdx, blockDim, gridDim;

#if !defined(__x86_64__)
#error "hipemu_switch is written for x86-64 (System V)"
#endif
// rbx, rbp, r12-r15 and the return address live on the stack being left; a new fiber's stack is laid out as if it had called this
asm(R"(
    .text
    .globl hipemu_switch
    .hidden hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
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
    .size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {
thread_local Runtime* tls_rt = nullptr;
int g_concurrent = 0;
int g_xcc_mode = 0;
unsigned g_xcc_start = 0;       // blocks launched so far (placement model 3, shim/hip/hip_runtime.h)

void trampoline() {
    Runtime& r = rt();
    r.body();
    r.done[r.cur] = true;
    --r.active;                                   // a returned thread no longer takes part in the block's barriers (producer waves leave early)
    if (r.active > 0 && r.blk_arrived >= r.active) { r.blk_arrived = 0; ++r.blk_gen; }
    hipemu_switch(&r.sp[r.cur], r.sched_sp);       // for good: a finished fiber is never switched to again
    abort();
}

// all threads of ONE block as fibers of the calling OS thread
static void run_block(dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz, const std::function<void()>& body, size_t stack_bytes) {
    Runtime& r = rt();
    const int n = (int)(block.x * block.y * block.z);
    for (int i = 0; i < n; ++i)
        if (!r.stack[i]) r.stack[i] = (char*)malloc(stack_bytes);
    r.body = body;
    r.nthreads = n;
    gridDim = Idx{grid.x, grid.y, grid.z};
    blockDim = Idx{block.x, block.y, block.z};
    r.blk_arrived = 0;
    r.active = n;
    for (auto& w : r.wave) w.arrived = 0;
    for (int i = 0; i < n; ++i) {
        void** top = (void**)(((uintptr_t)r.stack[i] + stack_bytes) & ~(uintptr_t)15);
        top[-1] = nullptr;                            // (where trampoline's caller would have left its return address: rsp = 8 mod 16 on entry)
        top[-2] = (void*)&trampoline;                 // hipemu_switch's `ret`
        for (int k = 3; k <= 8; ++k) top[-k] = nullptr;        // rbp, rbx, r12-r15
        r.sp[i] = (void*)(top - 8);
        r.done[i] = false;
        r.wait_word[i] = nullptr;
    }
    // HIPEMU_SCHEDULE: the order in which the scheduler visits the block's fibers - 0 (default) ascending, 1 descending, >= 2 a
    // seeded shuffle drawn anew on every pass.  Every order is a legal execution (threads of a block are unordered between
    // barriers; lanes of a wave meet at every cross-lane instruction), so results may differ between orders only in the
    // rounding of floating-point atomics - a kernel that reads LDS or global memory another wave writes WITHOUT a barrier in
    // between gives different (wrong) results under some order: tools/emu_schedules.sh runs the emulated tests under each.
    const char* sched_env = getenv("HIPEMU_SCHEDULE");
    const unsigned sched = sched_env ? (unsigned)atoi(sched_env) : 0u;
    unsigned long long lcg = 0x9E3779B97F4A7C15ull * (sched + 1u) + bx + 131u * by;
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = sched == 1 ? n - 1 - i : i;
    int left = n;
    while (left > 0) {
        if (sched >= 2)
            for (int i = n - 1; i > 0; --i) {
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                std::swap(order[i], order[(int)((lcg >> 33) % (unsigned)(i + 1))]);
            }
        for (int oi = 0; oi < n; ++oi) {
            const int i = order[oi];
            if (r.done[i]) continue;
            if (r.wait_word[i] && *r.wait_word[i] == r.wait_val[i]) continue;
            r.cur = i;
            blockIdx = Idx{bx, by, bz};
            threadIdx = Idx{(unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y)};
            hipemu_switch(&r.sched_sp, r.sp[i]);
            if (r.done[i]) --left;
        }
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int n = (int)(block.x * block.y * block.z);
    if (n > MAX_THREADS) { fprintf(stderr, "hipemu: %d threads per block\n", n); abort(); }
    const unsigned nblocks = grid.x * grid.y * grid.z;
    struct Advance { unsigned n; ~Advance() { g_xcc_start += n; } } advance{nblocks};
    if (g_concurrent && nblocks > 1) {
        if (nblocks > 256) { fprintf(stderr, "hipemu: %u concurrent blocks\n", nblocks); abort(); }
        std::vector<std::thread> ts;
        for (unsigned bz = 0; bz < grid.z; ++bz)
            for (unsigned by = 0; by < grid.y; ++by)
                for (unsigned bx = 0; bx < grid.x; ++bx)
                    ts.emplace_back([=, &body]() {
                        run_block(grid, block, bx, by, bz, body, 96 << 10);
                        Runtime& r = rt();                     // this OS thread ends with its block
                        for (int i = 0; i < MAX_THREADS; ++i) free(r.stack[i]);
                        delete tls_rt;
                        tls_rt = nullptr;
                    });
        for (auto& t : ts) t.join();
        return;
    }
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) run_block(grid, block, bx, by, bz, body, 1 << 20);
}
}  // namespace hipemu
extern "C" void hipemu_set_concurrent(int on) { hipemu::g_concurrent = on; }
extern "C" void hipemu_set_xcc_mode(int mode) { hipemu::g_xcc_mode = mode; }
