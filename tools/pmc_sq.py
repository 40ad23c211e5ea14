#!/usr/bin/env python3
"""Per-kernel SQ counters from several rocprofv3 --pmc passes (counters only, no trace domains) -> one table.

  python tools/pmc_sq.py <dir-with-pass-subdirs> [kernel-name-substring ...]

Every pass directory holds a *counter_collection.csv; counters are averaged per launch over the launches of a kernel name.
Units (MI355X_MICROARCH.md, cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD; SQ_LDS_IDX_ACTIVE = all LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra
cycles conflicts added (both per CU, summed over CUs); GRBM_GUI_ACTIVE = sum over the 8 XCDs of the launch's cycles.

Derived columns:
  conflict/active   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  lds_duty          = SQ_LDS_IDX_ACTIVE / (kernel cycles x CUs that ran blocks)    (kernel cycles = GRBM_GUI_ACTIVE / 8)
  mfma_duty         = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 4 SIMDs x CUs)
  wait_any, wait_inst_any, active_inst_any, wait_inst_lds : fractions of SQ_WAVE_CYCLES
"""
import csv
import glob
import json
import os
import sys


def load(root):
    acc = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].strip()
            if name.startswith("void "):
                name = name[5:]
            a = acc.setdefault(name, {})
            c = a.setdefault(r["Counter_Name"], [0, 0.0])
            c[0] += 1
            c[1] += float(r["Counter_Value"])
            g = a.setdefault("_grid", [0, 0.0])
            try:
                wg = float(r.get("Workgroup_Size", 0) or 0)
                gs = float(r.get("Grid_Size", 0) or 0)
                if wg > 0:
                    g[0] += 1
                    g[1] += gs / wg
            except ValueError:
                pass
    return acc


def main():
    root = sys.argv[1]
    pats = sys.argv[2:]
    ncu = int(os.environ.get("VPD_CUS", "256"))
    acc = load(root)
    rows = {}
    for k, a in acc.items():
        if pats and not any(p in k for p in pats):
            continue
        m = {c: v[1] / v[0] for c, v in a.items() if v[0]}
        if "GRBM_GUI_ACTIVE" not in m:
            continue
        blocks = m.get("_grid", ncu)
        cus = min(ncu, blocks) if blocks else ncu
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        d = {"launches": a["GRBM_GUI_ACTIVE"][0], "blocks": blocks, "kernel_cycles": cyc}
        lds_act, conf = m.get("SQ_LDS_IDX_ACTIVE"), m.get("SQ_LDS_BANK_CONFLICT")
        if lds_act:
            d["lds_idx_active"] = lds_act
            d["lds_duty"] = lds_act / (cyc * cus)
            if conf is not None:
                d["lds_bank_conflict"] = conf
                d["conflict_over_active"] = conf / lds_act
        for c in ("SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL", "SQ_INSTS_LDS",
                  "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_VMEM_RD",
                  "SQ_INSTS_VMEM_WR", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"):
            if c in m:
                d[c.lower()[3:]] = m[c]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            d["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"]
            d["mfma_duty"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 4.0 * cus)
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            d["wave_cycles"] = wc
            for c, n in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any"),
                         ("SQ_WAIT_INST_LDS", "wait_inst_lds")):
                if c in m:
                    d[n] = m[c] / wc
        rows[k] = d
    out = os.environ.get("PMC_SQ_JSON")
    if out:
        json.dump({"note": __doc__, "kernels": rows}, open(out, "w"), indent=1)
    hdr = ("kernel", "n", "blocks", "kcyc", "lds_duty", "confl/act", "mfma_duty", "wait_any", "wait_inst", "active", "w_i_lds")
    print("%-58s %4s %6s %8s %8s %9s %9s %8s %9s %7s %7s" % hdr)
    f = lambda d, k: ("%.3f" % d[k]) if k in d else "-"
    for k, d in sorted(rows.items(), key=lambda kv: -kv[1]["kernel_cycles"] * kv[1]["launches"]):
        print("%-58s %4d %6.0f %8.0f %8s %9s %9s %8s %9s %7s %7s" % (
            k[:58], d["launches"], d["blocks"], d["kernel_cycles"], f(d, "lds_duty"), f(d, "conflict_over_active"), f(d, "mfma_duty"),
            f(d, "wait_any"), f(d, "wait_inst_any"), f(d, "active_inst_any"), f(d, "wait_inst_lds")))
    print()
    print("raw per-launch means:")
    for k, d in sorted(rows.items(), key=lambda kv: -kv[1]["kernel_cycles"] * kv[1]["launches"]):
        print(k[:90])
        print("   " + "  ".join("%s=%.4g" % (n, v) for n, v in d.items() if n not in ("launches", "blocks")))


if __name__ == "__main__":
    main()
