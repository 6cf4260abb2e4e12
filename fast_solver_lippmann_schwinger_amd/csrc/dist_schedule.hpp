// Host arithmetic of the slab-distributed apply (dist.hip): how the owned x' range is cut into pipeline chunks, where the
// block of a (rank, chunk) pair sits in the message buffers, and which messages a rank posts -- in which order -- for one
// exchange.  Pure C++ (no HIP, no RCCL): dist.hip issues exactly this list through ncclSend / ncclRecv (or peer copies),
// tests/emu/dist_schedule_test.cpp checks it for P = 2, 4, 8 ranks without a GPU, and tests/test_distributed_cpu.py drives
// it over real gloo ranks.
//
// Buffers of rank r (complex elements; B = Wc * m * lz = one block):
//   S1 [P][K][B]  block (q, c) = what the x pass wrote for destination rank q, chunk c (way in: sent to q)
//                               = what came back from source rank q, chunk c          (way back: received from q)
//   R1 [K][P][B]  slot (c, q)  = block received from source rank q for chunk c        (way in)
//                               = z range of rank q of the transformed chunk c         (way back: sent to q)
// A block is [z][m][Wc] with z slowest, so its two z halves are its two contiguous halves (part 0 / 1).
#pragma once
#include <cstdint>
#include <vector>

namespace lsfc { namespace dsched {

struct Chunks { int W, K, Wc; };
// W = Lx / P storage indices of x' per rank; K <= requested chunks of at least one 8-wide tile each, K | W
inline Chunks plan_chunks(int Lx, int nranks, int requested) {
    Chunks ch; ch.W = Lx / nranks;
    int K = requested < 1 ? 1 : requested;
    while (K > 1 && (ch.W % K != 0 || (ch.W / K) % 8 != 0)) --K;
    ch.K = K; ch.Wc = ch.W / K;
    return ch;
}
inline int64_t s1_block(int q, int K, int c, int64_t B) { return ((int64_t)q * K + c) * B; }
inline int64_t r1_slot(int c, int P, int q, int64_t B) { return ((int64_t)c * P + q) * B; }

struct Msg {
    int peer;          // the other rank (== own rank: the local copy)
    bool send;         // send (read the buffer) or receive (write it)
    bool in_s1;        // the buffer: S1 or R1
    int64_t off, count;   // complex elements
};
// Messages of exchange (c, back, part) as rank `rank` posts them inside one group: first the local copy (as a send / receive
// pair with peer == rank), then for s = 1 .. P - 1 a send to rank + s and a receive from rank - s (pairwise schedule: in step s
// every rank talks to a different peer, so all links of a fully connected node carry traffic at once).
inline std::vector<Msg> exchange_messages(int rank, int P, int K, int c, bool back, int part, int64_t Bfull) {
    const int64_t B = part < 0 ? Bfull : Bfull / 2, off = part == 1 ? Bfull / 2 : 0;
    std::vector<Msg> out;
    auto s1 = [&](int q) { return s1_block(q, K, c, Bfull) + off; };
    auto r1 = [&](int q) { return r1_slot(c, P, q, Bfull) + off; };
    // way in: S1 block (q, c) -> rank q, lands in R1 slot (c, <source>); way back: R1 slot (c, q) -> rank q, lands in S1 block (<source>, c)
    out.push_back(Msg{rank, true, !back, back ? r1(rank) : s1(rank), B});
    out.push_back(Msg{rank, false, back, back ? s1(rank) : r1(rank), B});
    for (int s = 1; s < P; ++s) {
        const int to = (rank + s) % P, from = (rank - s + P) % P;
        out.push_back(Msg{to, true, !back, back ? r1(to) : s1(to), B});
        out.push_back(Msg{from, false, back, back ? s1(from) : r1(from), B});
    }
    return out;
}

}} // namespace lsfc::dsched
