#!/usr/bin/env python3
"""The tile-configuration model of csrc/igemm.hip (choose_cfg + plan_splits, restated here) against the measured sweeps
(tools/igemm_cfg_sweep.py tables): for every shape the configuration the model picks, what it costs against the best measured
one, and the total regret weighted by how often the shape occurs in one loop iteration.  CPU only.

    python tools/cfg_model_check.py profiles/r06/igemm_cfg_sweep_L_r06c.txt profiles/r06/igemm_cfg_sweep_M_r06c.txt [loop4=2150 epi3=19000 tiles128=1000000000 ...]
(the arguments in brackets restore round 5's constants: "shipped auto" in those tables was measured with them)
"""
import math, sys

OPTS = {  # bm, bn, slots, pro, loop, epi, epi_geglu, epi_side        (csrc/igemm.hip: choose_cfg)
    0: [256, 256, 256, 5000, 2650, 10500, 8700, 4000],
    1: [128, 320, 256, 3000, 2330, 9000, 9000, 4000],
    2: [128, 128, 512, 3000, 1900, 7000, 6000, 2000],
    3: [256, 320, 256, 5500, 3300, 16000, 11720, 9000],
    4: [128, 160, 512, 3000, 2430, 8000, 7000, 3000],
}
WAVE_W = {0: 128, 1: 80, 2: 64, 3: 160, 4: 160}
SPLIT = dict(min_nk=48, tiles128=300, max_nk_unsplit=80)
def choose(M, N, nk, geglu, side):
    best, bt = 2, 1e300
    for i, (bm, bn, slots, pro, loop, epi, epig, epis) in OPTS.items():
        if i == 1 and geglu:
            continue
        tiles = math.ceil(M / bm) * math.ceil(N / bn)
        rounds = math.ceil(tiles / slots)
        tile = pro + nk * loop + (epig if geglu else epi) + (epis if side else 0)
        ragged = 1.3 if (N > WAVE_W[i] and N % WAVE_W[i]) else 1.0
        t = rounds * tile * ragged
        if t < bt * 0.999:
            bt, best = t, i
    return best


def splits(M, N, nk, geglu):
    if geglu or N % 8:
        return 1
    tiles = math.ceil(M / 256) * math.ceil(N / 320)
    if tiles > 128 or nk < SPLIT["min_nk"]:
        return 1
    if math.ceil(M / 128) * math.ceil(N / 128) >= SPLIT["tiles128"] and nk <= SPLIT["max_nk_unsplit"]:
        return 1
    s = min(256 // tiles, nk // 6, 16)
    if s < 2:
        return 1
    per = math.ceil(nk / s)
    s = math.ceil(nk / per)
    return 1 if s < 2 else s


def main(argv):
    for a in argv:
        if "=" in a:
            k, v = a.split("=")
            if k in SPLIT:
                SPLIT[k] = int(v)
            else:                                                 # e.g. loop4=2430  epi3=19000
                name, c = k[:-1], int(k[-1])
                OPTS[c][["bm", "bn", "slots", "pro", "loop", "epi", "epig", "epis"].index(name)] = float(v)


    # launches per loop iteration of the swept shapes (profiles/r05/igemm_shapes_L_r05z4.txt; the same layers at the M row counts)
    COUNT = {(2560, 320): 0, (960, 320): 14, (320, 320): 17, (320, 1280): 0, (5120, 640): 21, (1920, 640): 14, (640, 640): 28, (640, 2560): 21,
             (10240, 1280): 21, (3840, 1280): 14, (1280, 1280): 28, (1280, 5120): 21, (1280, 11520): 12, (320, 2880): 11, (640, 5760): 9,
             (320, 960): 14, (640, 1920): 14, (1280, 3840): 14, (1280, 23040): 2, (640, 11520): 1, (320, 5760): 2, (1280, 2560): 2}
    tot_auto = tot_best = tot_model = 0.0
    for path in [a for a in argv if "=" not in a]:
        print(f"== {path}")
        for line in open(path):
            parts = [c.strip() for c in line.split("|")]
            head = parts[0].split()
            if len(head) != 5 or not head[0].isdigit():
                continue
            M, N, K, g, r = (int(v) for v in head)
            us = {}
            for c, cell in zip((0, 1, 2, 3, 4), parts[1:6]):
                if cell != "-":
                    us[c] = float(cell.split("us")[0])
            auto = float(parts[6].split("us")[0])
            nk = K // 64
            s = splits(M, N, nk, g)
            pick = 3 if s > 1 else choose(M, N, nk, g, r)
            # a forced configuration 3 in the sweep was measured WITH the split plan of the shipped build; an un-split pick of 3 where
            # the shipped build splits (or the reverse) is not in the table: flagged
            best = min(us, key=us.get)
            n = COUNT.get((N, K), 1)
            rows = M
            if rows in (4032, 1260):
                n = {(1280, 11520): 19, (10240, 1280): 6, (1280, 5120): 6, (1280, 3840): 22, (1280, 1280): 12, (3840, 1280): 4, (1280, 23040): 3,
                     (1280, 2560): 3}.get((N, K), 1)
            flag = "" if pick == best else f"   <- best {best} {us[best]:.1f}us ({100 * (us[pick] / us[best] - 1):+.0f} %)"
            print(f"{M:7d} {N:6d} {K:6d} g{g} r{r}  model {pick}{' split ' + str(s) if s > 1 else '':9s} {us[pick]:8.1f}us  auto(shipped) {auto:8.1f}us  x{n:2d}{flag}")
            tot_auto += n * auto; tot_best += n * us[best]; tot_model += n * us[pick]
    print(f"weighted per iteration: shipped auto {tot_auto / 1e3:.2f} ms, this model {tot_model / 1e3:.2f} ms, best measured {tot_best / 1e3:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1:])
