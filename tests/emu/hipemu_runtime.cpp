// TEST INFRASTRUCTURE: the fiber scheduler of tests/emu/shim/hip/hip_runtime.h (one definition per emulated library).
#include <hip/hip_runtime.h>

hipemu::Idx threadIdx, blockIdx, blockDim, gridDim;

namespace hipemu {
void trampoline() {
    Runtime& r = rt();
    r.body();
    r.done[r.cur] = true;
    --r.active;                                   // a returned thread no longer takes part in the block's barriers (producer waves leave early)
    if (r.active > 0 && r.blk_arrived >= r.active) { r.blk_arrived = 0; ++r.blk_gen; }
    swapcontext(&r.ctx[r.cur], &r.sched);
}
void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    Runtime& r = rt();
    const int n = (int)(block.x * block.y * block.z);
    if (n > MAX_THREADS) { fprintf(stderr, "hipemu: %d threads per block\n", n); abort(); }
    constexpr size_t STACK = 1 << 20;
    for (int i = 0; i < n; ++i)
        if (!r.stack[i]) r.stack[i] = (char*)malloc(STACK);
    r.body = body;
    r.nthreads = n;
    gridDim = Idx{grid.x, grid.y, grid.z};
    blockDim = Idx{block.x, block.y, block.z};
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                r.blk_arrived = 0;
                r.active = n;
                for (auto& w : r.wave) w.arrived = 0;
                for (int i = 0; i < n; ++i) {
                    getcontext(&r.ctx[i]);
                    r.ctx[i].uc_stack.ss_sp = r.stack[i];
                    r.ctx[i].uc_stack.ss_size = STACK;
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
}
}  // namespace hipemu
