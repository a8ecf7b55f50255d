// TEST INFRASTRUCTURE (CPU, g++; never linked into the product): a static "sanitizer" for the exchange protocol of the persistent
// GRU scans (pb_sed_amd/csrc/gru_stack.hip; reference op site pb_sed/models/weak_label/crnn.py:61-67).  The scans hand their step
// outputs from workgroup to workgroup as tagged words in a workspace ("the data is the flag"): a wrong index map does not crash,
// it times out or - worse - feeds a consumer the word of another row or unit.  This program compiles the SAME TEXT the kernels
// compile (gru_granule_map.h, gru_granule_role.inc) and enumerates it for one scan shape:
//   * every word a gate thread publishes is published by exactly one thread (no two writers);
//   * every (chain, layer, step, batch row < B, hidden unit < H) has a word (nothing a consumer waits for is never written);
//   * every word a contraction lane polls is the word of the (chain, source layer, source step, row, unit) its MFMA operand order
//     assumes - lane (lq, lr), load n, element e <-> batch row lr of the tile, unit k0 + 16 n + 4 lq + e, the order of its W fragment;
//   * the workgroups of a ring (chain, layer, batch tile) share block id % 8, i.e. one XCD / one L2 (the XCD-local BPTT exchange
//     publishes with plain stores and relies on it), and every XCD's share of the 1-D grid fits its CUs (co-residency);
//   * no index leaves the workspace the host allocates.
// Usage: scan_protocol_check fwd|bwd nchains nlayers B H T tiles_per_block cus      -> "OK ..." (exit 0) or "FAIL ..." (exit 1)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <tuple>
#include <vector>

#include "../pb_sed_amd/csrc/gru_granule_map.h"

using pbsed::gmap::Role;
using pbsed::gmap::RoleArgs;

static int fail(const char* fmt, long a = 0, long b = 0, long c = 0, long d = 0, long e = 0) {
    std::printf("FAIL ");
    std::printf(fmt, a, b, c, d, e);
    std::printf("\n");
    return 1;
}

struct Meaning { int chain, layer, t, row, unit; };
static bool operator==(const Meaning& x, const Meaning& y) {
    return x.chain == y.chain && x.layer == y.layer && x.t == y.t && x.row == y.row && x.unit == y.unit;
}

int main(int argc, char** argv) {
    if (argc != 9) return fail("usage: fwd|bwd nchains nlayers B H T tiles_per_block cus");
    const bool bwd = !std::strcmp(argv[1], "bwd");
    const int nchains = std::atoi(argv[2]), nlayers = std::atoi(argv[3]), B = std::atoi(argv[4]), H = std::atoi(argv[5]),
              T = std::atoi(argv[6]), NB = std::atoi(argv[7]), cus = std::atoi(argv[8]);
    if (!(H == 64 || H == 128 || H == 256 || H == 512) || (bwd && NB != 1) || NB < 1 || NB > 2) return fail("unsupported shape");
    const int KB = H == 64 ? 1 : H == 128 ? 1 : H == 256 ? 2 : 4, NW = H == 64 ? 4 : 8, NL = KB;     // gru_stack.hip: switch (H)
    if (KB * NW * 16 != H) return fail("KB * NW * 16 != H");
    const int nj = H / 16, nby = (B + 16 * NB - 1) / (16 * NB), top = nlayers - 1;
    const int Bp = (B + 15) / 16 * 16;
    const size_t per_cl = (size_t)T * Bp * H;
    const size_t ws_words = (size_t)nchains * nlayers * per_cl;      // the state part of the workspace (ops.py / pbsed.h)
    const int ngroups = nchains * (2 * nlayers - 1);
    // grid: 1-D XCD-aware mapping when every XCD can hold its share, else the 3-D grid (granule_xcd_grid)
    const int slots = pbsed::gmap::slots_1d(nchains, nlayers, nby, nj);
    const bool one_d = slots <= cus / 8;
    const int nblocks = one_d ? 8 * slots : nj * nby * ngroups;
    if (nblocks > cus) return fail("%ld blocks cannot be co-resident on %ld CUs (the launchers refuse this shape)", nblocks, cus);

    std::vector<Role> roles(nblocks);
    for (int blk = 0; blk < nblocks; ++blk) {
        if (one_d) {
            const RoleArgs ra{nj, nchains, nlayers, nby};
            roles[blk] = pbsed::gmap::role_1d((unsigned)blk, ra);
        } else {                                                     // gru_stack.hip::granule_role, 3-D branch
            const int x = blk % nj, y = (blk / nj) % nby, z = blk / (nj * nby);
            roles[blk] = Role{x, y, z % nchains, z / nchains, false};
        }
    }
    // 1. the role map is a bijection onto (chain, group, batch group, producer)
    std::set<std::tuple<int, int, int, int>> seen;
    std::map<std::tuple<int, int, int>, int> ring_xcd;              // (chain, gid, by) -> block id % 8
    for (int blk = 0; blk < nblocks; ++blk) {
        const Role& r = roles[blk];
        if (r.idle) continue;
        if (r.chain < 0 || r.chain >= nchains || r.gid < 0 || r.gid >= 2 * nlayers - 1 || r.by < 0 || r.by >= nby || r.bx < 0 || r.bx >= nj)
            return fail("block %ld: role out of range (chain %ld gid %ld by %ld bx %ld)", blk, r.chain, r.gid, r.by, r.bx);
        if (!seen.insert(std::make_tuple(r.chain, r.gid, r.by, r.bx)).second) return fail("block %ld repeats a role", blk);
        if (one_d && !(r.gid & 1)) {
            auto key = std::make_tuple(r.chain, r.gid, r.by);
            auto it = ring_xcd.find(key);
            if (it == ring_xcd.end()) ring_xcd[key] = blk & 7;
            else if (it->second != (blk & 7)) return fail("ring (chain %ld, group %ld, tile %ld) is spread over XCDs %ld and %ld", r.chain, r.gid, r.by, it->second, blk & 7);
        }
    }
    if ((long)seen.size() != (long)ngroups * nby * nj) return fail("%ld roles filled, %ld expected", (long)seen.size(), (long)ngroups * nby * nj);

    // 2. publishers: ring blocks, gate threads 0..255, rows < B
    std::map<size_t, Meaning> pub;
    for (int blk = 0; blk < nblocks; ++blk) {
        const Role& role = roles[blk];
        if (role.idle || (role.gid & 1)) continue;
        const int chain = role.chain;
        const int layer = bwd ? top - ((role.gid + 1) >> 1) : (role.gid + 1) >> 1;
        for (int t = 0; t < T; ++t)
            for (int nb = 0; nb < NB; ++nb)
                for (int tid = 0; tid < 256; ++tid) {
                    const int bb = (tid >> 4) & 15, u = tid & 15;
                    const int row = role.by * 16 * NB + nb * 16 + bb;
                    if (row >= B) continue;                            // bv[nb]
                    struct { int nlayers; } a = {nlayers};
                    size_t w;
                    if (bwd) w = 0 + PBSED_GM_RING_BASE(chain, a.nlayers, layer, per_cl, role.by, H, role.bx) + PBSED_GM_RING_WORD(t, Bp, H, tid);
                    else w = 0 + PBSED_GM_RING_BASE(chain, a.nlayers, layer, per_cl, role.by * NB, H, role.bx) + PBSED_GM_RING_WORD_NB(t, Bp, H, nb, tid);
                    if (w >= ws_words) return fail("block %ld publishes word %ld outside the workspace of %ld words", blk, (long)w, (long)ws_words);
                    const Meaning m{chain, layer, t, row, role.bx * 16 + u};
                    if (!pub.insert({w, m}).second) return fail("word %ld has two publishers (block %ld, thread %ld)", (long)w, blk, tid);
                }
    }
    // 3. completeness
    if (pub.size() != (size_t)nchains * nlayers * T * B * H)
        return fail("%ld words published, %ld = chains x layers x T x B x H expected", (long)pub.size(), (long)nchains * nlayers * T * B * H);

    // 4. consumers: contraction waves of rings (own previous step) and projections (neighbouring layer, same step)
    long polled = 0;
    const unsigned step_t = (unsigned)(Bp * H * 4);
    const unsigned tile_bytes = 16u * H * 4u;
    for (int blk = 0; blk < nblocks; ++blk) {
        const Role& role = roles[blk];
        if (role.idle) continue;
        const bool is_proj = role.gid & 1;
        const int chain = role.chain;
        const int layer = bwd ? top - ((role.gid + 1) >> 1) : (role.gid + 1) >> 1;
        const int src_layer = bwd ? (is_proj ? layer + 1 : layer) : (is_proj ? layer - 1 : layer);
        if (src_layer < 0 || src_layer >= nlayers) return fail("block %ld: source layer %ld", blk, src_layer);
        const unsigned cl_src = chain * nlayers + src_layer;
        for (int wave = 0; wave < NW; ++wave) {
            const int k0 = wave * NL * 16;
            for (int lane = 0; lane < 64; ++lane) {
                const int lq = lane >> 4, lr = lane & 15;
                const unsigned voff0 = bwd ? PBSED_GM_POLL_OFFSET0(cl_src, per_cl, role.by, H, k0, lr, lq)
                                           : PBSED_GM_POLL_OFFSET0(cl_src, per_cl, role.by * NB, H, k0, lr, lq);
                for (int step = 0; step < T; ++step) {
                    // scan order does not matter here: a ring reads its own words of the step before (any t with a predecessor),
                    // a projection the source ring's words of the same t
                    const int t_src = step;
                    for (int nb = 0; nb < NB; ++nb) {
                        const int row = role.by * 16 * NB + nb * 16 + lr;
                        if (row >= B) continue;                        // rowv[nb]: the lane's answer is not looked at
                        for (int n = 0; n < NL; ++n) {
                            const unsigned byte = voff0 + (unsigned)t_src * step_t + nb * tile_bytes + n * 1024;
                            if (byte & 15) return fail("block %ld: poll load at byte %ld is not 16-byte aligned", blk, byte);
                            for (int e = 0; e < 4; ++e) {
                                const size_t w = byte / 4 + e;
                                auto it = pub.find(w);
                                if (it == pub.end()) return fail("block %ld wave %ld lane %ld polls word %ld that nobody publishes", blk, wave, lane, (long)w);
                                const Meaning want{chain, src_layer, t_src, row, k0 + n * 16 + lq * 4 + e};
                                if (!(it->second == want))
                                    return fail("block %ld wave %ld lane %ld: polled word %ld is (row %ld, unit ..) of another element", blk, wave, lane, (long)w, it->second.row);
                                ++polled;
                            }
                        }
                    }
                }
            }
        }
    }
    std::printf("OK %s chains %d layers %d B %d H %d tiles/block %d: %d blocks (%s grid, %d per XCD of %d CUs), %zu words, %ld polled words checked\n",
                bwd ? "bwd" : "fwd", nchains, nlayers, B, H, NB, nblocks, one_d ? "1-D XCD-aware" : "3-D", one_d ? slots : 0, cus / 8, pub.size(), polled);
    return 0;
}
