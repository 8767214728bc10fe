#!/usr/bin/env python3
"""Regret of choose_cfg()'s cycle model on a measured sweep: for every row of an igemm_cfg_sweep file, the time of the
configuration the model picks against the best measured one.
    python tools/cost_model_check.py profiles/r01/igemm_cfg_sweep_v21.txt [more sweep files]
The constants below mirror posetraj_amd/csrc/igemm.hip (choose_cfg); keep them in step."""
import math, re, sys

OPTS = {  # cfg: (bm, bn, slots, pro, loop, epi, epi_geglu, epi_side)
    0: (256, 256, 256, 5000, 2650, 10500, 8700, 4000),
    1: (128, 320, 256, 3000, 2330, 9000, 9000, 4000),
    2: (128, 128, 512, 3000, 1900, 7000, 6000, 2000),
    3: (256, 320, 256, 5500, 3300, 19000, 12400, 9000),
    4: (128, 160, 512, 3000, 2150, 8000, 7000, 3000),
}
COLS = [0, 1, 2, 3, 4]

def model(M, N, K, geglu, side):
    best, bt = None, 1e300
    for c, (bm, bn, slots, pro, loop, epi, epig, epis) in OPTS.items():
        if c == 1 and geglu:
            continue
        tiles = math.ceil(M / bm) * math.ceil(N / bn)
        rounds = math.ceil(tiles / slots)
        t = rounds * (pro + (K // 64) * loop + (epig if geglu else epi) + (epis if side else 0))
        if t < bt * 0.999:
            best, bt = c, t
    return best

tot_pick = tot_best = 0.0
for fn in sys.argv[1:]:
    for line in open(fn):
        p = [x.strip() for x in line.split("|")]
        if len(p) < 6 or not p[0] or not p[0].split()[0].isdigit():
            continue
        M, N, K, g, r = (int(v) for v in p[0].split())
        times = {}
        for c, cell in zip(COLS, p[1:6]):
            m = re.match(r"([0-9.]+)us", cell)
            if m:
                times[c] = float(m.group(1))
        pick = model(M, N, K, bool(g), bool(r))
        best = min(times, key=times.get)
        tot_pick += times[pick]; tot_best += times[best]
        flag = "" if times[pick] <= times[best] * 1.03 else f"   <-- {100 * (times[pick] / times[best] - 1):.0f} % over"
        print(f"{M:7d} {N:6d} {K:6d} g{g} r{r}: model picks cfg {pick} ({times[pick]:.1f} us), best cfg {best} ({times[best]:.1f} us){flag}")
print(f"total: picked {tot_pick:.0f} us vs best {tot_best:.0f} us  (regret {100 * (tot_pick / tot_best - 1):.1f} %)")
