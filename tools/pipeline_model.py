"""Fluid model of the pipelined slab-distributed apply (csrc/dist.hip) on one rank: x pass -> K chunked exchanges in
(the link is shared with the exchanges back) -> per-chunk y/z/y passes -> exchanges back -> inverse x pass, with the
z-half split of the pipeline ends.  Inputs are the measured per-rank kernel times of
profiles/r03_dist_per_rank_compute_sim.log and the link model of DESIGN.md section 5; the output is what chunk count K
the schedule wants at a given link efficiency.  A model to choose defaults before the first multi-GPU run, nothing more.

usage: python tools/pipeline_model.py"""


def apply_ms(K, eta, link_full, comp_total, pen, x1, x3, dt=0.0005):
    """link_full: ms of one transpose at the nominal link rate; comp_total: y+z+y passes; pen: chunking penalty on them"""
    link = [link_full / eta / K] * K
    in_rem, back_rem, comp_rem = link[:], link[:], [comp_total * (1 + pen) / K] * K
    in_done, comp_done, back_done = [None] * K, [None] * K, [None] * K
    ci = cc = cb = 0
    t = 0.0
    while cb < K:
        act_in = ci < K and t >= (x1 / 2 if ci == 0 else x1)
        if ci == 0 and in_rem[0] <= link[0] / 2 and t < x1:          # second z half of chunk 0 needs the whole x pass
            act_in = False
        act_back = cb < K and comp_done[cb] is not None
        share = 0.5 if (act_in and act_back) else 1.0
        if act_in:
            in_rem[ci] -= dt * share
            if in_rem[ci] <= 0:
                in_done[ci] = t; ci += 1
        if act_back:
            back_rem[cb] -= dt * share
            if back_rem[cb] <= 0:
                back_done[cb] = t; cb += 1
        if cc < K and in_done[cc] is not None and t >= x1:
            comp_rem[cc] -= dt
            if comp_rem[cc] <= 0:
                comp_done[cc] = t; cc += 1
        t += dt
    return max(back_done[-1] + x3 / 2, back_done[-1] - link[-1] / 2 + x3)


if __name__ == "__main__":
    single = 12.33                                            # one GPU, round 3 (profiles/r03_bench_n512.json)
    # per rank at 512^3 (ms): x passes, y+z+y passes (round 3, simulated ranks with the receive buffers primed with real data:
    # profiles/r03_dist_per_rank_compute_sim.log), chunking penalty by K (round 1 measurement), one transpose at the nominal
    # per-direction link rate
    cases = {8: dict(x1=0.182, x3=0.209, comp_total=1.417, link_full=0.874, pen={2: -0.02, 4: 0.0, 8: 0.08, 16: 0.16}),
             4: dict(x1=0.380, x3=0.383, comp_total=2.576, link_full=3.495, pen={2: -0.01, 4: 0.0, 8: 0.05, 16: 0.10}),
             2: dict(x1=0.678, x3=0.760, comp_total=4.989, link_full=13.98, pen={2: 0.0, 4: 0.0, 8: 0.02, 16: 0.05})}
    for P, c in cases.items():
        for eta in (0.7, 0.8, 0.9, 1.0):
            row = {K: apply_ms(K, eta, c["link_full"], c["comp_total"], c["pen"][K], c["x1"], c["x3"]) for K in (2, 4, 8, 16)}
            best = min(row, key=row.get)
            print(f"P={P} link efficiency {eta:.1f}: " + "  ".join(f"K={K}: {v:6.2f} ms" for K, v in row.items())
                  + f"   -> K={best}, speed-up {single / row[best]:.2f}x")
