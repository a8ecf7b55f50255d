// TEST INFRASTRUCTURE: the fiber scheduler of tests/emu/shim/hip/hip_runtime.h (one definition per emulated library).
#include <hip/hip_runtime.h>

#include <thread>

thread_local hipemu::Idx threadIdx, blockIdx, blockDim, gridDim;

namespace hipemu {
thread_local Runtime* tls_rt = nullptr;
int g_concurrent = 0;

void trampoline() {
    Runtime& r = rt();
    r.body();
    r.done[r.cur] = true;
    --r.active;                                   // a returned thread no longer takes part in the block's barriers (producer waves leave early)
    if (r.active > 0 && r.blk_arrived >= r.active) { r.blk_arrived = 0; ++r.blk_gen; }
    swapcontext(&r.ctx[r.cur], &r.sched);
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
        getcontext(&r.ctx[i]);
        r.ctx[i].uc_stack.ss_sp = r.stack[i];
        r.ctx[i].uc_stack.ss_size = stack_bytes;
        r.ctx[i].uc_link = &r.sched;
        makecontext(&r.ctx[i], (void (*)())trampoline, 0);
        r.done[i] = false;
    }
    int left = n;
    while (left > 0) {
        for (int i = 0; i < n; ++i) {
            if (r.done[i]) continue;
            r.cur = i;
            blockIdx = Idx{bx, by, bz};
            threadIdx = Idx{(unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y)};
            swapcontext(&r.sched, &r.ctx[i]);
            if (r.done[i]) --left;
        }
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int n = (int)(block.x * block.y * block.z);
    if (n > MAX_THREADS) { fprintf(stderr, "hipemu: %d threads per block\n", n); abort(); }
    const unsigned nblocks = grid.x * grid.y * grid.z;
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
