#!/usr/bin/env python3
"""Golden vectors for `pantas quant` / `pantas remap`: runs the REAL reference scripts
(/root/reference/scripts/quantify.py, remap.py) in the build container on event CSVs of six synthetic replicates
(two conditions x three, same graph, different reads and expression) and stores inputs + outputs under
tests/golden/quant/.  Only data is stored; the reference source stays where it is.

    python tools/make_golden_quant.py [--ref /root/reference] [--out tests/golden/quant]
"""
import argparse
import gzip
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

QUANT_FLAGS = [[], ["--both"], ["-w", "3"], ["-w", "2", "--both"], ["-w", "-5"]]
REMAP_FLAGS = [[], ["-i", "1"], ["-i", "400"]]


def save_gz(path, text):
    """expected outputs are stored gzip-compressed (mtime 0: reproducible bytes)"""
    with open(path + ".gz", "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
        f.write(text.encode())


def run_ref(ref, script, args):
    return subprocess.run([sys.executable, os.path.join(ref, "scripts", script), *args], capture_output=True, text=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "quant"))
    a = ap.parse_args()
    import augment_oracle
    import call_oracle
    from pantas_amd import synth
    os.makedirs(a.out, exist_ok=True)
    tmp = os.path.join(a.out, "_tmp")
    os.makedirs(tmp, exist_ok=True)
    # one graph / annotation, six read sets: condition 2 expresses the genes differently (other Zipf order)
    base = dict(seed=5, n_genes=7, max_exons=7, bubble_rate=0.05, p_novel_skip=0.08, p_novel_ce=0.05, p_intron=0.12, p_novel_ss=0.06)
    names = []
    for cond in (1, 2):
        for rep in (1, 2, 3):
            pre = os.path.join(tmp, f"c{cond}_r{rep}")
            synth.write_all(pre, n_lines=2500 + 400 * rep, read_seed=100 * cond + rep, zipf_a=0.6 if cond == 1 else 1.6, **base)
            with open(pre + ".aug.gfa", "w") as f:
                augment_oracle.augment(pre + ".gaf", pre + ".gfa", f)
            rows = call_oracle.call(pre + ".aug.gfa", pre + ".gtf", None, rca=2 if rep != 3 else -1, novel=True, minintronsize=5)
            name = f"c{cond}_r{rep}.csv"
            with open(os.path.join(a.out, name), "w") as f:
                f.write(call_oracle.HEADER + "\n")
                for r in rows:
                    f.write(r + "\n")
            names.append(name)
            print(name, len(rows), "rows")
    os.replace(os.path.join(tmp, "c1_r1.gtf"), os.path.join(a.out, "anno.gtf"))
    c1 = [os.path.join(a.out, n) for n in names[:3]]
    c2 = [os.path.join(a.out, n) for n in names[3:]]
    meta = {"quant": [], "remap": [], "crashes": []}
    for i, flags in enumerate(QUANT_FLAGS):
        for nrep in ((3, 1) if i < 3 else (3,)):
            r = run_ref(a.ref, "quantify.py", [*flags, "--c1", *c1[:nrep], "--c2", *c2[:nrep]])
            assert r.returncode == 0, r.stderr
            name = f"quant_{i}_{nrep}.csv"
            save_gz(os.path.join(a.out, name), r.stdout)
            meta["quant"].append({"flags": flags, "replicates": nrep, "expected": name, "rows": len(r.stdout.splitlines()) - 1})
    # unequal replicate counts: a message on stdout, exit 0
    r = run_ref(a.ref, "quantify.py", ["--c1", *c1, "--c2", *c2[:2]])
    meta["unequal"] = {"stdout": r.stdout, "returncode": r.returncode}
    for src in ("quant_0_3.csv", "quant_2_1.csv"):
        for j, flags in enumerate(REMAP_FLAGS):
            plain = os.path.join(tmp, src)
            open(plain, "w").write(gzip.open(os.path.join(a.out, src + ".gz"), "rt").read())
            r = run_ref(a.ref, "remap.py", [*flags, plain, os.path.join(a.out, "anno.gtf")])
            assert r.returncode == 0, r.stderr
            name = f"remap_{src[:-4]}_{j}.csv"
            save_gz(os.path.join(a.out, name), r.stdout)
            meta["remap"].append({"flags": flags, "input": src, "expected": name, "rows": len(r.stdout.splitlines()) - 1})
    # inputs the scripts raise on
    bad = {
        "short_row.csv": call_oracle.HEADER + "\nES,annotated,chr1,g,+,a,b\n",
        "bad_cov.csv": call_oracle.HEADER + "\nES,annotated,chr1,g,+,T_R1.1.3,T_R1.1.2,T_R1.2.3,1>5,x,1>2,3,4>5,3\n",
        "bad_type.csv": call_oracle.HEADER + "\nXX,annotated,chr1,g,+,T_R1.1.3,T_R1.1.2,T_R1.2.3,1>5,2,1>2,3,4>5,3\n",
        "bad_nodes.csv": call_oracle.HEADER + "\nES,annotated,chr1,g,+,T_R1.1.3,T_R1.1.2,T_R1.2.3,1>a,2,1>2,3,4>5,3\n",
        "dot_nodes.csv": call_oracle.HEADER + "\nES,novel,chr1,g,+,?,T_R1.1.2,T_R1.2.3,.,2,1>2,3,4>5,3\n",
    }
    for name, text in bad.items():
        open(os.path.join(a.out, name), "w").write(text)
        r = run_ref(a.ref, "quantify.py", ["--c1", os.path.join(a.out, name), "--c2", c2[0]])
        assert r.returncode != 0, name
        meta["crashes"].append({"script": "quantify", "input": name, "exception": r.stderr.strip().splitlines()[-1].split(":")[0]})
    bad_remap = {
        "remap_short.csv": "ES,annotated,chr1,g,+,a,b\n",
        "remap_exon_range.csv": "ES,annotated,chr1,g,+,NOPE_R1.1.99,?,?,1>2,.,.,1/1,1/1,0.5,0.5,0.0\n",
    }
    for name, text in bad_remap.items():
        open(os.path.join(a.out, name), "w").write(text)
        r = run_ref(a.ref, "remap.py", [os.path.join(a.out, name), os.path.join(a.out, "anno.gtf")])
        assert r.returncode != 0, name
        meta["crashes"].append({"script": "remap", "input": name, "exception": r.stderr.strip().splitlines()[-1].split(":")[0]})
    json.dump(meta, open(os.path.join(a.out, "expected.json"), "w"), indent=1)
    import shutil
    shutil.rmtree(tmp)
    print(json.dumps({k: (len(v) if isinstance(v, list) else v) for k, v in meta.items()}))


if __name__ == "__main__":
    main()
