"""Derived per-kernel figures from two rocprofv3 passes of the same command (each `--kernel-trace --pmc ...`):
  pass A: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
          SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
  pass B: SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD
          SQ_WAIT_INST_LDS SQ_INSTS_SALU
usage: python tools/pmc_derived.py <dirA> <dirB> <kernel regex>
Columns: launches, us per launch (kernel trace of pass A), matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
duration x shader clock, the clock taken from SQ_BUSY_CYCLES / 32 shader engines / duration), non-MFMA vector / LDS /
scalar / vector-memory-read instructions per MFMA, LDS bank-conflict cycles per active LDS cycle, share of wave cycles
spent waiting (any / on an instruction's operands), vector-ALU and any-instruction issue activity as a share of the SIMD
cycles (SQ_ACTIVE_INST_* x 4; includes the MFMA issue slots)."""
import collections
import csv
import glob
import re
import sys


def load(d):
    cc, kt = glob.glob(d + "/*_counter_collection.csv")[0], glob.glob(d + "/*_kernel_trace.csv")[0]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))}
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return dur, per


def main():
    da, db, pat = sys.argv[1], sys.argv[2], re.compile(sys.argv[3])
    out = {}
    for which, d in (("a", da), ("b", db)):
        dur, per = load(d)
        for did, c in per.items():
            t, name = dur.get(did, (0, ""))
            if not pat.search(name):
                continue
            m = re.search(r"(\w+<[^>]*>)", name.replace("(anonymous namespace)::", ""))
            key = m.group(1) if m else name[:50]
            o = out.setdefault(key, {"a": collections.defaultdict(float), "b": collections.defaultdict(float), "na": 0, "nb": 0})
            o["n" + which] += 1
            o[which]["ns"] += t
            for k, v in c.items():
                o[which][k] += v
    print(f"{'kernel':34s} {'n':>4s} {'us':>8s} {'GHz':>5s} {'mfma busy':>9s} {'valu/mfma':>9s} {'lds/mfma':>8s} {'salu/mfma':>9s} "
          f"{'vmem/mfma':>9s} {'bankconf':>8s} {'wait':>5s} {'waitinst':>8s} {'valu busy':>9s} {'inst busy':>9s}")
    for key in sorted(out):
        a, b, na = out[key]["a"], out[key]["b"], max(out[key]["na"], 1)
        mf = a["SQ_INSTS_MFMA"]
        ghz = a["SQ_BUSY_CYCLES"] / 32.0 / max(a["ns"], 1)
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * a["ns"] * ghz) if ghz > 0 else 0.0
        scale = mf / max(mf, 1)  # both passes ran the same launches

        def per_mfma(x):
            return x * scale / mf if mf > 0 else float("nan")

        print(f"{key:34s} {na:4d} {a['ns'] / na / 1e3:8.1f} {ghz:5.2f} {busy * 100:8.1f}% {per_mfma(b['SQ_INSTS_VALU'] - mf):9.1f} "
              f"{per_mfma(b['SQ_INSTS_LDS']):8.2f} {per_mfma(b['SQ_INSTS_SALU']):9.2f} {per_mfma(b['SQ_INSTS_VMEM_RD']):9.2f} "
              f"{b['SQ_LDS_BANK_CONFLICT'] / max(b['SQ_LDS_IDX_ACTIVE'], 1):8.2f} {a['SQ_WAIT_ANY'] / max(a['SQ_WAVE_CYCLES'], 1):5.2f} "
              f"{a['SQ_WAIT_INST_ANY'] / max(a['SQ_WAVE_CYCLES'], 1):8.2f} "
              # SQ_ACTIVE_INST_* count quad-cycles of waves issuing that instruction class (x 4 = SIMD cycles)
              f"{a['SQ_ACTIVE_INST_VALU'] * 4 / (1024.0 * a['ns'] * ghz) * 100 if ghz > 0 else 0.0:8.1f}% "
              f"{a['SQ_ACTIVE_INST_ANY'] * 4 / (1024.0 * a['ns'] * ghz) * 100 if ghz > 0 else 0.0:8.1f}%")


if __name__ == "__main__":
    main()
