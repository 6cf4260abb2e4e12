// CPU check of the slab-exchange schedule that dist.hip posts through RCCL (csrc/dist_schedule.hpp), for P = 2, 4, 8 ranks,
// K = 1, 2, 4 chunks, both directions and the z-half split: every send has exactly one matching receive of equal size at the
// peer, in the same position of the (rank pair)'s message order; no two messages of an exchange overlap in a buffer; over
// all chunks (and halves) the messages tile the buffers exactly; in every step of the pairwise schedule each rank talks to a
// different peer.  Also the extern "C" surface the gloo test drives (built as a shared object with -DLSFC_SCHED_SHARED).
#include "../../fast_solver_lippmann_schwinger_amd/csrc/dist_schedule.hpp"
#include <algorithm>
#include <cstdio>
#include <map>
#include <set>
using namespace lsfc::dsched;

extern "C" {
int lsfc_sched_plan_chunks(int Lx, int nranks, int requested, int* W, int* K, int* Wc) {
    const Chunks c = plan_chunks(Lx, nranks, requested); *W = c.W; *K = c.K; *Wc = c.Wc; return 0;
}
// messages of exchange (c, back, part) on `rank`: peer / send / in_s1 / off / count per message; returns their number
int lsfc_sched_exchange(int rank, int P, int K, int c, int back, int part, int64_t Bfull, int cap, int* peer, int* send, int* in_s1, int64_t* off, int64_t* count) {
    const std::vector<Msg> m = exchange_messages(rank, P, K, c, back != 0, part, Bfull);
    if ((int)m.size() > cap) return -1;
    for (size_t i = 0; i < m.size(); ++i) { peer[i] = m[i].peer; send[i] = m[i].send; in_s1[i] = m[i].in_s1; off[i] = m[i].off; count[i] = m[i].count; }
    return (int)m.size();
}
}

#ifndef LSFC_SCHED_SHARED
static int bad = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++bad; if (bad < 20) { printf("FAIL: " __VA_ARGS__); printf("\n"); } } } while (0)

int main() {
    long checked = 0;
    for (int P : {1, 2, 4, 8}) for (int Kreq : {1, 2, 4, 8}) for (int Lx : {64, 96, 160, 1024, 1280, 1536}) {
        if ((Lx / 8) % P) continue;
        const Chunks ch = plan_chunks(Lx, P, Kreq);
        CHECK(ch.W * P == Lx && ch.K >= 1 && ch.K <= Kreq && ch.Wc * ch.K == ch.W && (ch.Wc % 8 == 0 || ch.K == 1), "plan_chunks(%d, %d, %d)", Lx, P, Kreq);
        const int K = ch.K;
        const int64_t B = (int64_t)ch.Wc * 6 * 4;               // m = 6, lz = 4 (even: z halves)
        for (int back = 0; back < 2; ++back) {
            // coverage of the buffers over all chunks: way in reads all of S1 and writes all of R1; way back the reverse
            for (int halves = 0; halves < 2; ++halves) {
                std::vector<std::vector<char>> rd((size_t)P, std::vector<char>((size_t)(P * K * B), 0)), wr = rd;
                for (int c = 0; c < K; ++c) for (int part : (halves ? std::vector<int>{0, 1} : std::vector<int>{-1})) {
                    std::vector<std::vector<Msg>> all((size_t)P);
                    for (int r = 0; r < P; ++r) all[(size_t)r] = exchange_messages(r, P, K, c, back != 0, part, B);
                    for (int r = 0; r < P; ++r) {
                        const auto& m = all[(size_t)r];
                        CHECK((int)m.size() == 2 * P, "message count");
                        CHECK(m[0].peer == r && m[0].send && m[1].peer == r && !m[1].send, "local copy first");
                        // pairwise steps: message 2s is the send to r + s, 2s + 1 the receive from r - s
                        std::set<int> to, from;
                        for (int s = 1; s < P; ++s) {
                            CHECK(m[(size_t)(2 * s)].send && m[(size_t)(2 * s)].peer == (r + s) % P, "send of step %d", s);
                            CHECK(!m[(size_t)(2 * s + 1)].send && m[(size_t)(2 * s + 1)].peer == (r - s + P) % P, "recv of step %d", s);
                            to.insert(m[(size_t)(2 * s)].peer); from.insert(m[(size_t)(2 * s + 1)].peer);
                        }
                        CHECK((int)to.size() == P - 1 && (int)from.size() == P - 1 && !to.count(r) && !from.count(r), "every peer exactly once");
                        for (const Msg& x : m) {
                            CHECK(x.count == (halves ? B / 2 : B), "message size");
                            CHECK(x.send ? (x.in_s1 == !back) : (x.in_s1 == (back != 0)), "buffer of a message");
                            auto& mark = x.send ? rd[(size_t)r] : wr[(size_t)r];
                            for (int64_t i = x.off; i < x.off + x.count; ++i) { CHECK(i >= 0 && i < (int64_t)mark.size() && !mark[(size_t)i], "overlap / out of range"); if (i >= 0 && i < (int64_t)mark.size()) mark[(size_t)i] = 1; }
                            ++checked;
                        }
                    }
                    // matching: the i-th send r -> q pairs with the i-th receive of q from r (one each per exchange), equal size,
                    // and the data lands where the layout says: way in block (q, c) of r's S1 -> slot (c, r) of q's R1
                    for (int r = 0; r < P; ++r) for (int q = 0; q < P; ++q) {
                        std::vector<Msg> snd, rcv;
                        for (const Msg& x : all[(size_t)r]) if (x.send && x.peer == q) snd.push_back(x);
                        for (const Msg& x : all[(size_t)q]) if (!x.send && x.peer == r) rcv.push_back(x);
                        CHECK(snd.size() == 1 && rcv.size() == 1 && snd[0].count == rcv[0].count, "send %d -> %d unmatched", r, q);
                        if (snd.size() == 1 && rcv.size() == 1) {
                            const int64_t ho = part == 1 ? B / 2 : 0;
                            CHECK(snd[0].off == (back ? r1_slot(c, P, q, B) : s1_block(q, K, c, B)) + ho, "send offset");
                            CHECK(rcv[0].off == (back ? s1_block(r, K, c, B) : r1_slot(c, P, r, B)) + ho, "recv offset");
                        }
                    }
                    // step s of the pairwise schedule: the sends of all ranks go to P different peers (all links busy at once)
                    for (int s = 1; s < P; ++s) { std::set<int> dst; for (int r = 0; r < P; ++r) dst.insert(all[(size_t)r][(size_t)(2 * s)].peer); CHECK((int)dst.size() == P, "step %d: a peer is addressed twice", s); }
                }
                for (int r = 0; r < P; ++r) {
                    CHECK(std::count(rd[(size_t)r].begin(), rd[(size_t)r].end(), 1) == (long)rd[(size_t)r].size(), "reads do not cover the source buffer (P=%d K=%d back=%d halves=%d)", P, K, back, halves);
                    CHECK(std::count(wr[(size_t)r].begin(), wr[(size_t)r].end(), 1) == (long)wr[(size_t)r].size(), "writes do not cover the destination buffer (P=%d K=%d back=%d halves=%d)", P, K, back, halves);
                }
            }
        }
    }
    printf("messages checked: %ld, failures: %d\n", checked, bad);
    return bad ? 1 : 0;
}
#endif
