#!/usr/bin/env python3
"""Fills the measured numbers of DESIGN.md from the round's profile files, so that the document's tables and the committed evidence cannot
drift apart: python tools/fill_design.py docs/DESIGN.template.md profiles r06 > DESIGN.md   (placeholders are @NAME@)."""
import json
import os
import sys


def last_json(path):
    for ln in reversed(open(path).read().strip().split("\n")):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit("no JSON line in " + path)


def main():
    tmpl, pdir, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
    P = lambda name: os.path.join(pdir, f"{rnd}_{name}")   # noqa: E731
    b = last_json(P("bench_final.json"))
    k = b["kernel_ms_per_step"]
    kc = b.get("kernel_ceilings") or {}

    def ceil(slot):
        e = kc.get(slot)
        if not e or "valu_frac" not in e:
            return "no counters for this source"
        s = f"**{e['bound'].upper()}** {e['frac_of_bound']:.2f}: VALU {e['valu_frac']:.2f} ({e['valu_instr_per_step']:.3g} + {e['salu_instr_per_step']:.3g} scalar wave-instructions per step)"
        if e.get("lds_pipe_frac") is not None:
            s += f", LDS pipe {e['lds_pipe_frac']:.2f} ({100 * (e.get('lds_conflict_share') or 0):.0f} % of it bank conflicts)"
        if e.get("hbm_frac") is not None:
            s += f", HBM {e['hbm_frac']:.2f}"
        return s
    s2 = b["roofline_stage2"]
    v = b["valu"]
    rep = {
        "WEIGHT_MS": f"{k['kmer_weight']:.2f}", "WEIGHT_CEIL": ceil("kmer_weight"),
        "MINHASH_MS": f"{k['minhash']:.1f}", "MINHASH_CEIL": ceil("minhash") + f"; {v['xorshift_steps_per_s']:.3g} chain steps/s = {v['frac_of_spec_ceiling']:.2f} of the spec-clock ceiling, {v['frac_of_measured_ceiling']:.2f} of the measured-clock one",
        "ORDERED_MS": f"{k['ordered']:.2f} (both launches, the weighted MinHash launch inside the first's time)", "ORDERED_CEIL": ceil("ordered"),
        "IBUILD_MS": f"{k['index_build']:.2f}", "IBUILD_CEIL": ceil("index_build"),
        "IQUERY_MS": f"{k['index_query']:.2f}", "IQUERY_CEIL": ceil("index_query"),
        "JOIN_MS": f"{k['overlap']:.2f}", "STAGE2": f"{s2['frac']:.2f} of HBM peak ({s2['achieved'] / 1e3:.2f} TB/s algorithmic, {s2['pairs_per_s'] / 1e6:.0f} M pairs/s); " + ceil("overlap"),
        "STEP": f"**{b['ms_per_step']:.1f} ms → {b['value'] / 1e3:.1f} k overlaps/s, {b['sketches_per_sec'] / 1e6:.2f} M sketches/s** (`{rnd}_bench_final.json`: 20 steps; soak {b['soak']['steps']} steps, mean {b['soak']['mean_ms_per_step']:.1f})",
        "SEARCH_WALL": f"{b['phase_wall_ms']['search']:.1f} ms",
        "E2E": f"{b['end_to_end']['wall_s']:.2f} s",
        "CPU": f"{b['cpu_baseline']['value']:.0f} overlaps/s", "RATIO": f"{b['value'] / b['cpu_baseline']['value']:.0f}",
    }
    d1 = b.get("dist_path_1rank") or {}
    rep["DIST1"] = (f"{d1['ms_per_step']:.1f} ms against {b['ms_per_step']:.1f}: the eager path runs the ordered kernel first — {d1['kernel_ms_per_step']['ordered']:.1f} ms — and packs / gathers on top"
                    if "ms_per_step" in d1 else "not measured")
    c1, c4s, c4, c5s, c5r = (last_json(P(f"bench_{c}.json")) for c in ("c1", "c4slice", "c4", "c5slice", "c5rank"))
    rep["C1"] = f"{c1['ms_per_step']:.2f} ms/step"
    rep["C4IQ"] = f"{c4['kernel_ms_per_step']['index_query']:.1f}"
    rep["C4"] = f"{c4s['ms_per_step']:.1f} ms / {c4['ms_per_step'] / 1e3:.3f} s"
    rep["C5"] = f"{c5s['ms_per_step']:.0f} ms / {c5r['ms_per_step'] / 1e3:.2f} s ({c5r['value'] / 1e6:.1f} M overlaps/s, {c5r['records_per_step'] / 1e6:.1f} M records)"
    ranks = [json.loads(ln) for ln in open(P("emulate_rank.txt")) if ln.startswith("{")]
    rows = ["| N | config | reads per rank | sketch + index ms | search of ALL queries ms | rank step ms (no exchange) | one-GPU step | bound |", "|---|---|---|---|---|---|---|---|"]
    one = {"c2": b["ms_per_step"], "c4": c4["ms_per_step"], "c5": None}
    r8 = {}
    seen = set()
    for r in ranks:
        o = one.get(r["config"])
        dup = (r["world"], r["config"]) in seen      # (the second N = 8 line at C2: the eager add's kernel order, MHAP_ORDERED_FIRST=1)
        seen.add((r["world"], r["config"]))
        if dup:
            rows.append(f"| {r['world']} (ordered kernel first: the eager add's order) | {r['config']} | {r['reads_per_rank']} | {r['sketch_and_index_ms']:.1f} | {r['search_all_queries_ms']:.1f} | **{r['rank_step_ms_without_comm']:.1f}** | "
                        + (f"{o:.1f} | {o / r['rank_step_ms_without_comm']:.2f}× |" if o else "— | — |"))
            continue
        rows.append(f"| {r['world']} | {r['config']} | {r['reads_per_rank']} | {r['sketch_and_index_ms']:.1f} | {r['search_all_queries_ms']:.1f} | **{r['rank_step_ms_without_comm']:.1f}** | "
                    + (f"{o:.1f} | {o / r['rank_step_ms_without_comm']:.2f}× |" if o else "— | — |"))
        if r["world"] == 8:
            r8[r["config"]] = r
    rep["RANKTABLE"] = "\n".join(rows)
    rep["C5HBM"] = f"{r8['c5']['hbm_gb']['in_use']:.0f}" if "c5" in r8 else "?"
    rep["RANK8"] = "C2 %.1f ms, C4 %.0f ms, C5 %.1f s" % (r8["c2"]["rank_step_ms_without_comm"], r8["c4"]["rank_step_ms_without_comm"], r8["c5"]["rank_step_ms_without_comm"] / 1e3)
    model = r8["c2"]["rank_step_ms_without_comm"] + 0.4
    rep["R8C2"] = f"{r8['c2']['rank_step_ms_without_comm']:.1f}"
    rep["SPEEDUP_SHORT"] = f"{b['ms_per_step'] / model:.1f}× (N = 4: {b['ms_per_step'] / (next(r for r in ranks if r['world'] == 4 and r['config'] == 'c2')['rank_step_ms_without_comm'] + 0.3):.2f}×)"
    rep["STAGE2_SHORT"] = f"{s2['frac']:.2f} of HBM peak, {k['overlap']:.2f} ms at C2"
    smi = (b.get("soak") or {}).get("smi") or {}
    if (b.get("soak") or {}).get("smi_ok") and smi.get("sclk_MHz_hwmon") and (smi.get("power_W_input") or smi.get("power_W_average")):
        pw = smi.get("power_W_input") or smi.get("power_W_average")
        rep["SMI"] = (f"{pw['mean']:.0f} W on average against the {smi['power_cap_W']['mean']:.0f}-W cap (max {pw['max']:.0f}) at a mean shader clock of "
                      f"{smi['sclk_MHz_hwmon']['mean']:.0f} MHz ({smi['sclk_MHz_hwmon']['min']:.0f}-{smi['sclk_MHz_hwmon']['max']:.0f}) over the soak leg, {pw['n']} samples")
    else:
        rep["SMI"] = "the capture of this round FAILED (no clock or power samples on the box): soak.smi_ok false"
    eager = [r for r in ranks if r.get("world") == 8 and r["config"] == "c2"]
    rep["R8C2EAGER"] = f"{eager[-1]['rank_step_ms_without_comm']:.1f}" if len(eager) > 1 else "not measured"
    rep["N8_OVERLAPS"] = f"{b['records_per_step'] / ((r8['c2']['rank_step_ms_without_comm'] + 0.4) * 1e-3) / 1e6:.2f} M"
    rep["N1_OVERLAPS"] = f"{b['value'] / 1e6:.3f} M"
    eager_part = (f"; with the eager add's kernel order, which is what a real run executes: {b['ms_per_step']:.1f} / ({eager[-1]['rank_step_ms_without_comm']:.1f} + 0.4) = **{b['ms_per_step'] / (eager[-1]['rank_step_ms_without_comm'] + 0.4):.1f}×**"
                  if len(eager) > 1 else "")
    rep["SPEEDUP"] = f"{b['ms_per_step']:.1f} / ({r8['c2']['rank_step_ms_without_comm']:.1f} + 0.4 exposed) = **{b['ms_per_step'] / model:.1f}×**{eager_part} (a model until a real run; C4: {one['c4'] / r8['c4']['rank_step_ms_without_comm']:.1f}× before the exchange)"
    text = open(tmpl).read()
    for inc in ("LIMITS", "SWITCHES", "LAYOUT"):
        pth = os.path.join(os.path.dirname(tmpl), f"DESIGN.{inc.lower()}.md")
        if os.path.exists(pth):
            rep[inc] = open(pth).read().rstrip("\n")
    extra = os.path.join(os.path.dirname(tmpl), "DESIGN.values.json")     # hand-entered figures (ranges over boxes, A/B results)
    if os.path.exists(extra):
        rep.update(json.load(open(extra)))
    for name, val in rep.items():
        text = text.replace("@" + name + "@", str(val))
    left = sorted({w for w in __import__("re").findall(r"@[A-Z0-9_]+@", text)})
    if left:
        sys.stderr.write("unfilled placeholders: %s\n" % left)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
