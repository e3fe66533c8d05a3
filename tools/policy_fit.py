#!/usr/bin/env python3
"""policy_fit.py -- fit MMH_KERNEL_AUTO's cost table to a measured dataset and judge it on held-out shapes (no GPU).

    python tools/policy_fit.py --fit gpurun_out/r04/dataset_fit.json [--heldout gpurun_out/r04/dataset_heldout.json]
                               [--emit how-to-optimize-gemm_amd/csrc/policy_table.inc] [--report profiles/r04_auto_regret.md]

The dataset (tools/tile_sweep.py over tools/policy_shapes_*.txt) holds, per shape, the TFLOP/s of every tile family
forced as a plain launch (/sk0) and as a persistent stream-K launch (/sk2).  The model AUTO evaluates per candidate
(csrc/policy.hip, the same arithmetic as predict() below):

    plain:     t = fix_p + cmax * (nk * s_p[o] + tile_p[o])           cmax = ceil(tiles / CUs): tiles on the fullest CU, o = min(cmax, w)
    stream-K:  t = fix_s[w'] + (tiles / CUs) * (nk * s_s[w'] + tile_s[w'])   w' = persistent workgroups per CU

with nk = ceil(k / 32) K-slices per tile and w the family's co-residency; fix_p and fix_s[w'] come in two flavours: the
guarded instantiation's (any m, n, k) and the whole-tile one's (m, n multiples of the tile, k of 32, 16-byte rows:
csrc/internal.hpp fast_shape) -- no bounds tests in front of the first DMA and around the C stores: 1-2 us less on a
plain launch, 5-7 us less on a persistent one (N = 1152 / 1280: 8.5 us against the 13.4 of N = 1151 / 1279); a plain launch of more than one round of
workgroups whose last round is not full (cmax > w, tiles not a multiple of w CUs) is priced `margin` times its prediction -- the 90th percentile of measured / predicted over the fit
set's multi-round plain rows: which CU gets the last tiles, and when, is the dispatcher's business, and the residuals
of those rows are one-sided.  s_x[o] is what a CU takes per tile-slice
with o tiles co-resident, tile_x[o] what it takes per tile besides (pipeline fill, C store, a stream-K part's hand-over),
fix_x what a launch costs whatever its size.
All of it is per family, in microseconds, least squares in relative error over the fit set's rows of that family
and form.  Everything is expressed per CU, so the table serves any CU count (partitioned devices).

This replaces round 3's hand-set thresholds (the reference's `NEW := MMult_xxx` choice, cuda/makefile:1-3)."""
from __future__ import annotations

import argparse
import json
import re
import math
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUS = 256
THIN = 0.55    # what a round of K2W's thin edge tiles costs, in rounds of whole tiles (N = 1025 against 1024, plain: 26.3 against 18.1 us)
PAIRING = 1.10   # a plain launch whose last round holds between half a tile and one tile per CU (cus / 2 < tiles mod (w cus) <= cus, more than
                 # one round): the dispatcher may hand those tiles out one per CU -- or in pairs, to the CUs whose two workgroups ended
                 # together, and the launch takes a whole extra round (round 6: 4822 x 1268 x 2551 on the 128x64 tile, 760 tiles,
                 # 142.2 TFLOP/s in one pass and 110.1 in the next; eight such shapes of 500).  Priced at the risk, not the median.
T64SK = 1.03     # persistent stream-K launches of the K2W 64x64 tile at two workgroups per CU (from 512 tiles): the one candidate whose rate moves from run to run (3000^3 .. 3500^3,
                 # N + 1 shapes: 141.9 / 136.8 / 143.8 / 137.7 TFLOP/s in the dataset passes, 139.6 - 141.7 / 133.1 - 135.2 / 138.6 - 141.4 / 133.4 - 134.0
                 # in five later runs of the same launches, 134.7 - 135.8 / 129.5 - 131.6 / 136.2 - 137.4 / 131.8 - 133.6 inside two off-grid
                 # passes; every other family repeats within 1 %) -- priced at the risk: with the fitted cost alone AUTO took it on 36 of
                 # 117 off-grid shapes and lost 5-8 % on a dozen of them (profiles/r06_notes.md section 10)
RIM5 = False     # MMH_OPT_RIM5 (tools build; measured, it loses): the fused rim launch of the 64x64 tile
# family: (kernel short name, BM, BN, co-resident workgroups per CU of a plain launch, has a stream-K form,
#          persistent workgroups per CU of its stream-K launches, the kernel id's macro)
# skw: launch_streamk bounds every stream-K grid of a tile by its GUARDED (chained) instantiation's residency -- K2W's
# 64x64 / 128x64 tiles need 116 / 165 registers there: two / one workgroup(s) per CU where three / two rings fit
# (tests/test_kernel_resources.py reads it off the binary); round 5 measured the whole-tile instantiations on their own,
# larger, residency (tools build, option 103): no gain anywhere on the sweep (profiles/r05_notes.md).  K2L's kernels have
# 256-thread workgroups and fit three / two / one.
FAMILIES = {
    "t64": ("mfma_64x64_dma5", 64, 64, 3, True, 2, "MMH_KERNEL_MFMA_64X64_DMA5"),
    "t128x64": ("mfma_128x64_dma5", 128, 64, 2, True, 1, "MMH_KERNEL_MFMA_128X64_DMA5"),
    "t128": ("mfma_128x128_dma5", 128, 128, 1, True, 1, "MMH_KERNEL_MFMA_128X128_DMA5"),
    "t96": ("mfma_96x96_dma5", 96, 96, 2, False, 0, "MMH_KERNEL_MFMA_96X96_DMA5"),
    "t96x64": ("mfma_96x64_dma5", 96, 64, 2, False, 0, "MMH_KERNEL_MFMA_96X64_DMA5"),
    "t160": ("mfma_160x160_dma5", 160, 160, 1, False, 0, "MMH_KERNEL_MFMA_160X160_DMA5"),   # round 6: ships since the fragment reads are spread (N = 2560: 256 tiles)
    "t256": ("mfma_256x256", 256, 256, 1, True, 1, "MMH_KERNEL_MFMA_256X256"),
    # K2L (round 3's LDS-DMA tiles, every wave issuing its share of the DMA): candidates since round 5
    "l64": ("mfma_64x64_dma", 64, 64, 2, False, 0, "MMH_KERNEL_MFMA_64X64_DMA"),   # plain launches only: its forced stream-K launches run three per CU (768 workgroups), a grid the two-per-CU model cannot price and AUTO would not launch
    "l128x64": ("mfma_128x64_dma", 128, 64, 2, True, 2, "MMH_KERNEL_MFMA_128X64_DMA"),
    "l128": ("mfma_128x128_dma", 128, 128, 1, True, 1, "MMH_KERNEL_MFMA_128X128_DMA"),
}


def tail_split(tiles, w, cus, k):
    """csrc/internal.hpp dma5_tail_split: the plain K2W launches that go out as one whole round + the last round as a launch of its own."""
    rem = tiles - w * cus
    return w >= 2 and k >= 512 and 100 * rem > 85 * cus and rem <= cus


def rim_dims(m, n):
    """csrc/internal.hpp dma5_rim_dims: rim rows / columns of the 64x64 tile's RIM launch (0, 0: none)."""
    rm, rn = m % 64, n % 64
    a = 1 if rm == 1 and m > 64 else 0
    b = 1 if rn == 1 and n > 64 else 0
    return a, b


def geometry(fam, m, n, k, cus=CUS):
    _, bm, bn, w, has_sk, skw, _macro = FAMILIES[fam]
    rim = False
    if fam == "t64" and RIM5:
        a, b = rim_dims(m, n)
        if a or b:        # the rim costs no tiles of its own, and runs as a plain launch only
            m, n, rim, has_sk = m - a, n - b, True, False
    nbm, nbn = -(-m // bm), -(-n // bn)
    tiles = nbm * nbn
    nk = -(-k // 32)
    cmax = -(-tiles // cus)
    # K2W's thin edge tiles (a last tile row / column of at most 16 valid rows / columns: a fraction of a tile's MFMAs,
    # dispatched last, beside whole tiles): a plain launch is priced as the whole tiles' rounds + THIN x the thin tiles'
    thin = 0
    if FAMILIES[fam][0].endswith("_dma5"):
        tr = 1 if nbm > 1 and m - (nbm - 1) * bm <= 16 else 0
        tc = 1 if nbn > 1 and n - (nbn - 1) * bn <= 16 else 0
        thin = tiles - (nbm - tr) * (nbn - tc)
    cfull = -(-(tiles - thin) // cus)
    cmax_p = cfull + THIN * (cmax - cfull)     # (thin tiles cost only where they add a tile to the fullest CU)
    wp = 0
    for cand in range(skw, 0, -1):      # (the grid launch_streamk launches: at most skw workgroups per CU)
        if tiles >= cand * cus:
            wp = cand
            break
    sk_possible = has_sk and wp > 0 and tiles % (wp * cus) != 0
    whole = m % bm == 0 and n % bn == 0 and k % 32 == 0      # dense operands, 16-byte aligned bases (how the sets are measured)
    return dict(tiles=tiles, nk=nk, cmax=cmax, cmax_p=cmax_p, occ=min(cmax, w), wp=wp, sk_possible=sk_possible, w=w, whole=whole)


def rows_of(dataset):
    """(family, form, shape, microseconds) for every measured candidate whose launch took the form it was asked for."""
    out = []
    for r in dataset:
        m, n, k = r["m"], r["n"], r["k"]
        flops = 2.0 * m * n * k
        for fam, (kern, *_rest) in FAMILIES.items():
            for suffix, form in (("/sk0", "plain"), ("/sk2", "sk"), ("", "plain")):
                key = kern + suffix
                tf = r.get(key)
                if not tf:
                    continue
                grid = 0
                if "forms" in r:      # compacted dataset (--compact): p[grid] = persistent, 1 = plain, x = another family ran
                    code = r["forms"].get(key, "x")
                    is_sk, own = code.startswith("p"), code != "x"
                    grid = int(code[1:]) if len(code) > 1 and code[1:].isdigit() else 0
                else:
                    launched = r.get("launched", {}).get(key, "")
                    is_sk = "persistent" in launched
                    own = "LDS-DMA" in launched or fam == "t256"
                    mm = re.search(r"on (\d+) persistent", launched)
                    grid = int(mm.group(1)) if mm else 0
                if (form == "sk") != is_sk:
                    continue
                if not own:
                    continue      # fell back to a register-staged tile (descriptor window): not this family
                if is_sk and grid and grid != geometry(fam, m, n, k)["wp"] * CUS:
                    continue      # (a grid the model does not describe: another residency than skw)
                out.append((fam, form, (m, n, k), flops / tf / 1e6))
    return out


def compact(dataset):
    """The dataset without its launch strings: per measured variant one letter (p = persistent launch, 1 = one
    workgroup per tile, x = the family did not take the shape) -- what is committed under profiles/."""
    out = []
    for r in dataset:
        c = {k: v for k, v in r.items() if k not in ("launched", "bit_equal_to_first")}
        forms = {}
        for key, text in r.get("launched", {}).items():
            fam_ok = "LDS-DMA" in text or key.startswith("mfma_256x256") or key == "auto"
            mm = re.search(r"on (\d+) persistent", text)
            forms[key] = "x" if not fam_ok else ("p" + (mm.group(1) if mm else "") if "persistent" in text else "1")
        c["forms"] = forms
        fa = family_of_auto(r)
        c["auto_is"] = list(fa) if fa else None      # (family, form) the pass's own MMH_KERNEL_AUTO launched
        out.append(c)
    return out


def fit(rows):
    table = {}
    for fam, (_, bm, bn, w, has_sk, skw, _macro) in FAMILIES.items():
        entry = {"bm": bm, "bn": bn, "w": w, "skw": skw, "fix_p": 0.0, "s_p": [0.0] * 3, "fix_s": [0.0] * 3, "s_s": [0.0] * 3,
                 "fix_p_whole": 0.0, "fix_s_whole": [0.0] * 3, "tile_p": [0.0] * 3, "tile_s": [0.0] * 3,
                 "n_p": 0, "n_s": 0, "rms_p": 0.0, "rms_s": 0.0}
        for form in ("plain", "sk"):
            sel = [(s, us) for (f, fo, s, us) in rows if f == fam and fo == form]
            if len(sel) < 4:
                continue
            # columns: fixed cost of the guarded instantiation (plain: one; stream-K: one per w' -- the hand-over of a
            # persistent workgroup costs more the more of them share a CU), the same of the whole-tile instantiation,
            # then the per-tile-slice time and the per-tile time per occupancy (shared by both instantiations)
            X, y = [], []
            for (m, n, k), us in sel:
                g = geometry(fam, m, n, k)
                feat = [0.0] * 12
                lo = 3 if g["whole"] else 0
                if form == "plain":
                    feat[lo] = 1.0
                    feat[6 + g["occ"] - 1] = g["cmax_p"] * g["nk"]
                    feat[9 + g["occ"] - 1] = g["cmax_p"]
                else:
                    feat[lo + g["wp"] - 1] = 1.0
                    feat[6 + g["wp"] - 1] = g["tiles"] * g["nk"] / CUS
                    feat[9 + g["wp"] - 1] = g["tiles"] / CUS
                X.append([f / us for f in feat])      # relative error
                y.append(1.0)
            X, y = np.array(X), np.array(y)
            used = [j for j in range(12) if np.any(X[:, j] != 0)]
            sol, *_ = np.linalg.lstsq(X[:, used], y, rcond=None)
            coef = [0.0] * 12
            for j, v in zip(used, sol):
                coef[j] = max(float(v), 0.0)
            # occupancies never observed inherit the nearest observed one; a whole-tile cost never observed, the guarded one
            for lo in (0, 3, 6, 9):
                for j in range(lo, lo + 3):
                    if coef[j] == 0.0 and not (lo < 6 and form == "plain"):
                        near = [coef[i] for i in (j - 1, j + 1, j - 2, j + 2) if lo <= i < lo + 3 and coef[i] > 0]
                        coef[j] = near[0] if near else 0.0
            for j in range(3):
                if coef[3 + j] == 0.0:
                    coef[3 + j] = coef[j]
            res = X @ np.array(coef) - y
            if form == "plain":
                entry.update(fix_p=coef[0], fix_p_whole=coef[3], s_p=coef[6:9], tile_p=coef[9:], n_p=len(sel),
                             rms_p=float(np.sqrt(np.mean(res ** 2))))
            else:
                entry.update(fix_s=coef[:3], fix_s_whole=coef[3:6], s_s=coef[6:9], tile_s=coef[9:], n_s=len(sel),
                             rms_s=float(np.sqrt(np.mean(res ** 2))))
        table[fam] = entry
    # multi-round plain launches: their residuals are one-sided (the tail of the last round) -- price them at the 90th
    # percentile of measured / predicted
    ratios = []
    for (f, fo, (m, n, k), us) in rows:
        g = geometry(f, m, n, k)
        if fo == "plain" and g["cmax"] > g["w"] and g["tiles"] % (g["w"] * CUS) != 0 and table[f]["n_p"]:
            e = table[f]
            ratios.append(us / (e["fix_p_whole" if g["whole"] else "fix_p"] +
                                g["cmax_p"] * (g["nk"] * e["s_p"][g["occ"] - 1] + e["tile_p"][g["occ"] - 1])))
    table["_margin"] = round(float(np.percentile(ratios, 90)), 3) if ratios else 1.0
    return table


def predict(table, fam, form, m, n, k, cus=CUS):
    e = table.get(fam)
    if not e or not e["n_p"]:
        return math.inf       # a family the dataset does not hold: not a candidate
    g = geometry(fam, m, n, k, cus)
    if form == "plain":
        t = e["fix_p_whole" if g["whole"] else "fix_p"] + g["cmax_p"] * (g["nk"] * e["s_p"][g["occ"] - 1] + e["tile_p"][g["occ"] - 1])
        rem = g["tiles"] % (g["w"] * cus)
        split = FAMILIES[fam][0].endswith("_dma5") and tail_split(g["tiles"], g["w"], cus, k)      # (those go out as two launches: not of the class)
        if g["cmax"] > g["w"] and cus / 2 < rem <= cus and not split:
            return t * max(PAIRING, table.get("_margin", 1.0))
        return t * table.get("_margin", 1.0) if g["cmax"] > g["w"] and rem != 0 else t
    if not g["sk_possible"] or e["n_s"] == 0:
        return math.inf
    return ((e["fix_s_whole" if g["whole"] else "fix_s"][g["wp"] - 1] +
             g["tiles"] / cus * (g["nk"] * e["s_s"][g["wp"] - 1] + e["tile_s"][g["wp"] - 1])) * (T64SK if fam == "t64" and g["wp"] >= 2 else 1.0))      # (two persistent workgroups per CU: from 512 tiles)


def choose(table, m, n, k, cus=CUS):
    best, best_t = None, math.inf
    for fam in FAMILIES:
        for form in ("plain", "sk"):
            t = predict(table, fam, form, m, n, k, cus)
            if t < best_t:
                best, best_t = (fam, form), t
    return best, best_t


def launch_of(r, key):
    """('plain' | 'sk' | None when another family ran, persistent grid or 0) of a measured variant."""
    if "forms" in r:
        code = r["forms"].get(key, "x")
        if code == "x":
            return None, 0
        return ("sk" if code.startswith("p") else "plain"), (int(code[1:]) if code[1:].isdigit() else 0)
    text = r.get("launched", {}).get(key, "")
    if not text or text.startswith("error"):
        return None, 0
    mm = re.search(r"on (\d+) persistent", text)
    return ("sk" if "persistent" in text else "plain"), (int(mm.group(1)) if mm else 0)


def family_of_auto(r):
    """(family, form) MMH_KERNEL_AUTO launched in this pass, from its launch string (None for compacted datasets)."""
    if "auto_is" in r:
        return tuple(r["auto_is"]) if r["auto_is"] else None
    text = r.get("launched", {}).get("auto", "")
    mm = re.search(r"<(\d+),(\d+)>", text)
    if not mm:
        return None
    k2w = "loader wave" in text
    for fam, (kern, bm, bn, *_rest) in FAMILIES.items():
        if (bm, bn) == (int(mm.group(1)), int(mm.group(2))) and (kern.endswith("_dma5") == k2w or fam == "t256") and \
                (fam != "t256" or "LDS-DMA" not in text) and (fam == "t256" or "LDS-DMA" in text):
            return fam, ("sk" if "persistent" in text else "plain")
    return None


def measured(r, fam, form):
    """TFLOP/s of the family forced in that launch form -- only if the launch TOOK that form (a forced /sk2 on a count
    the grid divides, or with too few whole tiles, runs plain: that is a second measurement of the plain launch)."""
    kern = FAMILIES[fam][0]
    for key in ([kern + "/sk2"] if form == "sk" else [kern + "/sk0", kern + "/sk2", kern]):
        if r.get(key) and launch_of(r, key)[0] == form:
            return r[key]
    if r.get("auto") and family_of_auto(r) == (fam, form):
        return r["auto"]      # (the pass's own AUTO launched exactly this candidate: e.g. stream-K over thin edge tiles)
    return None


def regret(table, dataset):
    out = []
    for r in dataset:
        m, n, k = r["m"], r["n"], r["k"]
        (fam, form), _ = choose(table, m, n, k)
        got = measured(r, fam, form)
        if got is None and form == "sk":
            # the table prices a persistent launch the pass did not measure (forced kernels decide plain / persistent on
            # the whole tiles alone when a last tile row / column is thin): judged as the plain launch of the family
            got = measured(r, fam, "plain")
        cands = {}
        for f in FAMILIES:
            for fo in ("plain", "sk"):
                v = measured(r, f, fo)
                if v:
                    cands[(f, fo)] = v
        best = max(cands.values())
        out.append(dict(shape=(m, n, k), chosen=f"{fam}/{form}", tf=got, best=best, best_is=max(cands, key=cands.get),
                        regret=1.0 - (got or 0.0) / best, old_auto=r.get("auto")))
    return out


def emit(table, path, source):
    with open(path, "w") as f:
        f.write("// policy_table.inc -- GENERATED by tools/policy_fit.py from " + source + "; do not edit by hand.\n")
        f.write("// Per tile family: co-residency w, has-stream-K, stream-K residency skw, then microseconds: plain launches t = fix_p + cmax * (nk * s_p[o] + tile_p[o]), o = min(cmax, w) - 1,\n")
        f.write("// persistent stream-K launches t = fix_s[w' - 1] + tiles / CUs * (nk * s_s[w' - 1] + tile_s[w' - 1]) (tools/policy_fit.py has the derivation;\n")
        f.write("// rows / rms relative residual of each fit behind it); then fix_p and fix_s of the whole-tile instantiation (shapes\n")
        f.write("// csrc/internal.hpp fast_shape accepts for the family's tile), then tile_p and tile_s.\n")
        fams = {k: v for k, v in table.items() if not k.startswith("_") and v["n_p"]}
        for fam, e in fams.items():
            f.write(f"// {fam}: plain {e['n_p']} rows, rms {e['rms_p']:.3f}; stream-K {e['n_s']} rows, rms {e['rms_s']:.3f}\n")
        f.write(f"#define MMH_POLICY_THIN {THIN:.2f}f   // a round of K2W's thin edge tiles, in rounds of whole tiles (plain launches)\n")
        f.write(f"#define MMH_POLICY_MULTIROUND_MARGIN {table.get('_margin', 1.0):.3f}f   // plain launches of more than one round: p90 of measured / predicted\n")
        f.write(f"#define MMH_POLICY_PAIRING_MARGIN {PAIRING:.3f}f   // ... whose last round holds between half a tile and one tile per CU: may pair up on half the CUs\n")
        f.write(f"#define MMH_POLICY_T64SK_MARGIN {T64SK:.3f}f   // stream-K launches of the K2W 64x64 tile: the one candidate whose rate moves from run to run\n")
        f.write("#define MMH_POLICY_FAMILIES \\\n")
        for fam, e in fams.items():
            kern = FAMILIES[fam][6]
            sp = ", ".join(f"{v:.6f}f" for v in e["s_p"])
            ss = ", ".join(f"{v:.6f}f" for v in e["s_s"])
            fs = ", ".join(f"{v:.4f}f" for v in e["fix_s"])
            fw = ", ".join(f"{v:.4f}f" for v in e["fix_s_whole"])
            tp = ", ".join(f"{v:.4f}f" for v in e["tile_p"])
            ts = ", ".join(f"{v:.4f}f" for v in e["tile_s"])
            f.write(f"  {{{kern}, {e['bm']}, {e['bn']}, {e['w']}, {1 if e['n_s'] else 0}, {e['skw'] if e['n_s'] else 0}, {e['fix_p']:.4f}f, {{{sp}}}, {{{fs}}}, {{{ss}}}, "
                    f"{e['fix_p_whole']:.4f}f, {{{fw}}}, {{{tp}}}, {{{ts}}}}}, \\\n")
        f.write("\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fit", required=True)
    ap.add_argument("--heldout", default="")
    ap.add_argument("--emit", default="")
    ap.add_argument("--report", default="")
    ap.add_argument("--families", default="", help="comma-separated subset of the families to fit and price (default: all the dataset holds)")
    ap.add_argument("--compact", default="", help="write the fit (and held-out) dataset without launch strings: PREFIX_fit.json, PREFIX_heldout.json")
    args = ap.parse_args()
    if args.families:
        for fam in list(FAMILIES):
            if fam not in args.families.split(","):
                del FAMILIES[fam]
    fit_set = json.load(open(args.fit))
    if args.compact:
        json.dump(compact(fit_set), open(args.compact + "_fit.json", "w"), separators=(",", ":"))
        if args.heldout:
            json.dump(compact(json.load(open(args.heldout))), open(args.compact + "_heldout.json", "w"), separators=(",", ":"))
    table = fit(rows_of(fit_set))
    print("multi-round plain margin", table["_margin"])
    for fam, e in table.items():
        if fam.startswith("_"):
            continue
        print(fam, json.dumps({k: (round(v, 4) if isinstance(v, float) else [round(x, 5) for x in v] if isinstance(v, list) else v)
                               for k, v in e.items()}))
    lines = []
    for label, path in (("fit", args.fit), ("held-out", args.heldout)):
        if not path:
            continue
        rg = regret(table, json.load(open(path)))
        rs = np.array([x["regret"] for x in rg])
        old = np.array([1.0 - (x["old_auto"] or 0.0) / x["best"] for x in rg])
        summary = (f"{label}: {len(rg)} shapes; regret of the table's choice against the best measured candidate: mean {rs.mean() * 100:.2f} %, "
                   f"p90 {np.percentile(rs, 90) * 100:.2f} %, max {rs.max() * 100:.2f} %; the `auto` column of the same pass (MMH_KERNEL_AUTO as the library "
                   f"stood when the set was measured -- round 3's rules in the first pass of round 4, the previous table in the second; it "
                   f"carries its own measurement noise): mean {old.mean() * 100:.2f} %, max {old.max() * 100:.2f} %")
        print(summary)
        lines.append(summary)
        worst = sorted(rg, key=lambda x: -x["regret"])[:12]
        for x in worst:
            lines.append(f"  {x['shape']}: chose {x['chosen']} {x['tf']} TF, best {x['best_is'][0]}/{x['best_is'][1]} {x['best']} ({x['regret'] * 100:.1f} %)")
        for l in lines[-12:]:
            print(l)
    if args.emit:
        emit(table, args.emit, os.path.basename(args.fit))
        print("wrote", args.emit)
    if args.report:
        with open(args.report, "w") as f:
            f.write("# MMH_KERNEL_AUTO: fitted cost table and its regret (tools/policy_fit.py)\n\n```\n")
            f.write(f"multi-round plain margin {table['_margin']}\n")
            for fam, e in table.items():
                if fam.startswith("_"):
                    continue
                f.write(fam + " " + json.dumps({k: (round(v, 4) if isinstance(v, float) else [round(x, 5) for x in v] if isinstance(v, list) else v)
                                                for k, v in e.items()}) + "\n")
            f.write("```\n\n" + "\n".join(lines) + "\n")
        print("wrote", args.report)


if __name__ == "__main__":
    main()
