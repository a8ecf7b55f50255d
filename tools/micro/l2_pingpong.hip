// Microbenchmark: hand-off latency between two workgroups through memory for different cache-scope bits, on the same
// XCD (block ids equal mod 8) and across XCDs.  Block A publishes i, block B waits for it and answers with i, A waits
// for the answer: time per round trip, and whether the value becomes visible at all (bounded spin).
//   aux bits of the gfx940+ buffer instructions: sc0 = 1, nt = 2, sc1 = 16.
// Build: hipcc -O3 --offload-arch=gfx950 l2_pingpong.hip -o l2_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ST, int LD>
__global__ void pingpong(unsigned* buf, int partner_a, int partner_b, int iters, unsigned* out) {
    const int me = blockIdx.x;
    if (threadIdx.x != 0) return;
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);        // HW_REG_XCC_ID[3:0]
    out[16 + me] = xcc & 15u;
    if (me != partner_a && me != partner_b) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 4096, 0x00020000);
    const bool is_a = me == partner_a;
    const unsigned mine = is_a ? 0u : 256u, theirs = is_a ? 256u : 0u;        // byte offsets (different lines)
    unsigned fails = 0;
    long long t0 = wall_clock64();
    for (int i = 1; i <= iters; ++i) {
        if (is_a) __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, mine, 0, ST);
        int spin = 0;
        for (;; ++spin) {
            asm volatile("" ::: "memory");        // the load builtin is not volatile: keep it inside the loop
            unsigned v;
            if (LD == 100) v = __hip_atomic_fetch_or(buf + theirs / 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (LD == 101) v = __hip_atomic_fetch_or(buf + theirs / 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (LD == 102) { asm volatile("buffer_inv sc0" ::: "memory"); v = __builtin_amdgcn_raw_buffer_load_b32(rs, theirs, 0, 0); }
            else if (LD == 103) { asm volatile("buffer_inv sc1" ::: "memory"); v = __builtin_amdgcn_raw_buffer_load_b32(rs, theirs, 0, 0); }
            else if (LD == 104) { asm volatile("buffer_inv sc0 sc1" ::: "memory"); v = __builtin_amdgcn_raw_buffer_load_b32(rs, theirs, 0, 0); }
            else v = __builtin_amdgcn_raw_buffer_load_b32(rs, theirs, 0, LD < 100 ? LD : 0);
            if (v >= (unsigned)i) break;
            if (spin > 2000000) { ++fails; break; }
        }
        if (fails) break;
        if (!is_a) __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, mine, 0, ST);
    }
    long long t1 = wall_clock64();
    if (is_a) { out[0] = (unsigned)(t1 - t0); out[1] = fails; }
    else out[2] = fails;
}

// Fresh address per iteration (the scans' situation: every exchanged word has its own location, written once per call): the
// FIRST look at a location may be a plain load - the CU's L1 cannot hold a stale copy of a line it never read, and inside an
// XCD the L2 is the coherence point - with sc1 loads only for the retries.  WAIT = clocks the consumer sleeps before the first
// look (0 = immediately: the first plain look usually comes too early and leaves the stale line in L1, every retry is sc1).
template <int ST, int FIRST, int WAIT>
__global__ void pingpong_fresh(unsigned* buf, int partner_a, int partner_b, int iters, unsigned* out) {
    const int me = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (me != partner_a && me != partner_b) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1u << 26, 0x00020000);
    const bool is_a = me == partner_a;
    unsigned fails = 0, first_ok = 0;
    long long t0 = wall_clock64();
    for (int i = 1; i <= iters; ++i) {
        const unsigned mine = (unsigned)i * 512u + (is_a ? 0u : 256u), theirs = (unsigned)i * 512u + (is_a ? 256u : 0u);
        if (is_a) __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, mine, 0, ST);
        for (int w = 0; w < WAIT; w += 64) __builtin_amdgcn_s_sleep(1);
        int spin = 0;
        for (;; ++spin) {
            asm volatile("" ::: "memory");
            const unsigned v = spin == 0 ? __builtin_amdgcn_raw_buffer_load_b32(rs, theirs, 0, FIRST)
                                         : __builtin_amdgcn_raw_buffer_load_b32(rs, theirs, 0, 16);
            if (v == (unsigned)i) { first_ok += spin == 0; break; }
            if (spin > 2000000) { ++fails; break; }
        }
        if (fails) break;
        if (!is_a) __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, mine, 0, ST);
    }
    long long t1 = wall_clock64();
    if (is_a) { out[0] = (unsigned)(t1 - t0); out[1] = fails; out[3] = first_ok; }
    else { out[2] = fails; out[4] = first_ok; }
}

// As pingpong_fresh<plain store, plain first look>, plus a WRITE-THROUGH (sc1) store of the same value to a second location
// EXTRA bytes further on, issued right behind the plain one (the scans publish a state twice: plain for the ring on this XCD,
// write-through for the readers on other XCDs).  Does the write-through store hold up the plain hand-off - and does it matter
// whether both lines map to the same L2 channel (EXTRA a multiple of 4 KB) or not (EXTRA + 256 B ...)?
template <int EXTRA, int WAIT>
__global__ void pingpong_dual(unsigned* buf, int partner_a, int partner_b, int iters, unsigned* out) {
    const int me = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (me != partner_a && me != partner_b) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1u << 26, 0x00020000);
    const bool is_a = me == partner_a;
    unsigned fails = 0, first_ok = 0;
    long long t0 = wall_clock64();
    for (int i = 1; i <= iters; ++i) {
        const unsigned mine = (unsigned)i * 512u + (is_a ? 0u : 256u), theirs = (unsigned)i * 512u + (is_a ? 256u : 0u);
        if (is_a) {
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, mine, 0, 0);
            if (EXTRA) __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, mine + (unsigned)EXTRA, 0, 16);
        }
        for (int w = 0; w < WAIT; w += 64) __builtin_amdgcn_s_sleep(1);
        int spin = 0;
        for (;; ++spin) {
            asm volatile("" ::: "memory");
            const unsigned v = spin == 0 ? __builtin_amdgcn_raw_buffer_load_b32(rs, theirs, 0, 0)
                                         : (EXTRA ? __builtin_amdgcn_raw_buffer_load_b32(rs, theirs + (unsigned)EXTRA, 0, 16)
                                                  : __builtin_amdgcn_raw_buffer_load_b32(rs, theirs, 0, 16));
            if (v == (unsigned)i) { first_ok += spin == 0; break; }
            if (spin > 2000000) { ++fails; break; }
        }
        if (fails) break;
        if (!is_a) {
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, mine, 0, 0);
            if (EXTRA) __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, mine + (unsigned)EXTRA, 0, 16);
        }
    }
    long long t1 = wall_clock64();
    if (is_a) { out[0] = (unsigned)(t1 - t0); out[1] = fails; out[3] = first_ok; }
    else { out[2] = fails; out[4] = first_ok; }
}

template <int EXTRA, int WAIT>
void run_dual(const char* name, unsigned* big, unsigned* out, int a, int b) {
    const int iters = 20000;
    hipMemset(big, 0, 1u << 26); hipMemset(out, 0, 256);
    hipLaunchKernelGGL((pingpong_dual<EXTRA, WAIT>), dim3(16), dim3(64), 0, 0, big, a, b, iters, out);
    hipDeviceSynchronize();
    unsigned h[32]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    printf("dual store, %-44s wait %4d blocks %2d<->%2d: %s  %.0f ns per round trip, first look ok %u / %u of %d\n", name, WAIT, a, b,
           (h[1] || h[2]) ? "NOT VISIBLE" : "ok", h[0] * 1e6 / rate / iters, h[3], h[4], iters);
}

template <int ST, int FIRST, int WAIT>
void run_fresh(const char* name, unsigned* big, unsigned* out, int a, int b) {
    const int iters = 20000;
    hipMemset(big, 0, 1u << 26); hipMemset(out, 0, 256);
    hipLaunchKernelGGL((pingpong_fresh<ST, FIRST, WAIT>), dim3(16), dim3(64), 0, 0, big, a, b, iters, out);
    hipDeviceSynchronize();
    unsigned h[32]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    printf("fresh address, %-34s wait %4d blocks %2d<->%2d: %s  %.0f ns per round trip, first look ok %u / %u of %d\n", name, WAIT, a, b,
           (h[1] || h[2]) ? "NOT VISIBLE" : "ok", h[0] * 1e6 / rate / iters, h[3], h[4], iters);
}

template <int ST, int LD>
void run(const char* name, unsigned* buf, unsigned* out, int a, int b) {
    const int iters = 20000;
    hipMemset(buf, 0, 4096); hipMemset(out, 0, 256);
    hipLaunchKernelGGL((pingpong<ST, LD>), dim3(16), dim3(64), 0, 0, buf, a, b, iters, out);
    hipDeviceSynchronize();
    unsigned h[32]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);     // kHz
    printf("%-28s blocks %2d<->%2d (xcc %u, %u): %s  %.0f ns per round trip\n", name, a, b, h[16 + a], h[16 + b],
           (h[1] || h[2]) ? "NOT VISIBLE (spin limit)" : "ok", (h[1] || h[2]) ? 0.0 : h[0] * 1e6 / rate / iters);
}

int main() {
    unsigned *buf, *out; hipMalloc(&buf, 4096); hipMalloc(&out, 256);
    for (int pass = 0; pass < 2; ++pass) {
        const int a = 0, b = pass == 0 ? 8 : 1;      // same XCD (ids equal mod 8) / neighbouring XCDs
        run<16, 16>("store sc1, load sc1", buf, out, a, b);
        run<16, 1>("store sc1, load sc0", buf, out, a, b);
        run<17, 1>("store sc0+sc1, load sc0", buf, out, a, b);
        run<1, 1>("store sc0, load sc0", buf, out, a, b);
        run<0, 1>("store plain, load sc0", buf, out, a, b);
        run<16, 3>("store sc1, load sc0+nt", buf, out, a, b);
        run<17, 17>("store sc0+sc1, load sc0+sc1", buf, out, a, b);
        run<16, 17>("store sc1, load sc0+sc1", buf, out, a, b);
        run<17, 16>("store sc0+sc1, load sc1", buf, out, a, b);
        run<19, 19>("store/load sc0+sc1+nt", buf, out, a, b);
        run<16, 100>("store sc1, load = atomic or(agent)", buf, out, a, b);
        run<16, 101>("store sc1, load = atomic or(wg)", buf, out, a, b);
        run<0, 101>("store plain, load = atomic or(wg)", buf, out, a, b);
        run<0, 102>("store plain, inv sc0 + plain load", buf, out, a, b);
        run<1, 102>("store sc0, inv sc0 + plain load", buf, out, a, b);
        run<16, 102>("store sc1, inv sc0 + plain load", buf, out, a, b);
        run<0, 103>("store plain, inv sc1 + plain load", buf, out, a, b);
        run<16, 103>("store sc1, inv sc1 + plain load", buf, out, a, b);
        run<16, 104>("store sc1, inv sc0sc1 + plain load", buf, out, a, b);
    }
    unsigned* big; hipMalloc(&big, 1u << 26);
    for (int pass = 0; pass < 2; ++pass) {
        const int a = 0, b = pass == 0 ? 8 : 1;
        run_fresh<16, 16, 0>("store sc1, looks sc1", big, out, a, b);
        run_fresh<16, 0, 0>("store sc1, first look plain", big, out, a, b);
        run_fresh<16, 0, 512>("store sc1, first look plain", big, out, a, b);
        run_fresh<16, 0, 768>("store sc1, first look plain", big, out, a, b);
        run_fresh<16, 0, 1024>("store sc1, first look plain", big, out, a, b);
        run_fresh<16, 0, 1536>("store sc1, first look plain", big, out, a, b);
        run_fresh<16, 16, 768>("store sc1, looks sc1", big, out, a, b);
        run_fresh<16, 16, 1024>("store sc1, looks sc1", big, out, a, b);
        run_fresh<16, 16, 1536>("store sc1, looks sc1", big, out, a, b);
        run_fresh<0, 0, 512>("store plain, first look plain", big, out, a, b);
        run_fresh<0, 0, 768>("store plain, first look plain", big, out, a, b);
        run_fresh<0, 0, 1024>("store plain, first look plain", big, out, a, b);
        if (pass == 0) {
            run_dual<0, 512>("plain store only", big, out, a, b);
            run_dual<(1 << 25), 512>("+ sc1 store 32 MB on (same channel)", big, out, a, b);
            run_dual<(1 << 25) + 256, 512>("+ sc1 store 32 MB + 256 B on", big, out, a, b);
            run_dual<(1 << 25) + 1024, 512>("+ sc1 store 32 MB + 1 KB on", big, out, a, b);
            run_dual<(1 << 25) + 4096 + 2048, 512>("+ sc1 store 32 MB + 6 KB on", big, out, a, b);
            run_dual<0, 768>("plain store only", big, out, a, b);
            run_dual<(1 << 25), 768>("+ sc1 store 32 MB on (same channel)", big, out, a, b);
            run_dual<(1 << 25) + 256, 768>("+ sc1 store 32 MB + 256 B on", big, out, a, b);
        }
        run_fresh<1, 0, 768>("store sc0, first look plain", big, out, a, b);
        run_fresh<16, 1, 768>("store sc1, first look sc0", big, out, a, b);
    }
    return 0;
}
