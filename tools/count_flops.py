#!/usr/bin/env python3
"""Developer tool: exact FP64 operation counts of the mode-N kernels, read off the gfx950 ISA hipcc generates (not an
estimate): compiles ccal_kernels_fused.hip to assembly, finds the corner loop of the Gram kernel (the innermost loop
that holds the v_fma_f64 stream) and counts, per lane and iteration = per corner,
    v_fma_f64 / v_fmac_f64 (2 flop), v_mul_f64 / v_add_f64 (1 flop), ds_add_f64 (1 flop, executed by the LDS),
and the other FP64 VALU instructions that occupy the same issue slots (rcp / rsq seeds, ldexp, conversions).  The
per-frame elimination kernel (k_schur1m) has no loop: its whole body is counted per lane (16 lanes per frame).
Writes profiles/<tag>/flops.json, which bench.py reads for extra.mode_N_roofline.

    python tools/count_flops.py r02
"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "camera_intrinsic_calibration_rs_amd", "csrc")
FLOP = {"v_fma_f64": 2, "v_fmac_f64_e32": 2, "v_fmac_f64_e64": 2, "v_mul_f64": 1, "v_add_f64": 1, "ds_add_f64": 1,
        "v_mul_f64_e32": 1, "v_add_f64_e32": 1, "v_fma_f64_e64": 2, "v_mul_f64_e64": 1, "v_add_f64_e64": 1}


def kernel_bodies(asm):
    lines = asm.split("\n")
    out = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN4ccal\w+):\s", lines[i])
        if m:
            j = i
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            out[m.group(1)] = lines[i:j]
            i = j
        i += 1
    return out


def ops(lines):
    c = collections.Counter()
    for l in lines:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    return c


def loops(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    res = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if labels.get(t, 1 << 30) < i:
                res.append((labels[t], i))
    return res


def summarize(c):
    f64 = {k: v for k, v in c.items() if "f64" in k and (k.startswith("v_") or k.startswith("ds_add"))}
    flops = sum(v * FLOP.get(k, 0) for k, v in f64.items())
    return {"flops": flops, "fma": sum(v for k, v in f64.items() if "fma" in k),
            "mul_add": sum(v for k, v in f64.items() if FLOP.get(k) == 1 and k.startswith("v_")),
            "lds_add_f64": c.get("ds_add_f64", 0),
            "f64_valu_instructions": sum(v for k, v in f64.items() if k.startswith("v_")),
            "all_instructions": sum(c.values()),
            "valu_instructions": sum(v for k, v in c.items() if k.startswith("v_")),
            "lds_instructions": sum(v for k, v in c.items() if k.startswith("ds_")),
            # register-file traffic that is not arithmetic: AGPR <-> VGPR copies (accumulators that do not fit the 256 registers
            # the VALU addresses), the row trade of k_gram2, scratch
            "v_accvgpr_copies": sum(v for k, v in c.items() if "accvgpr" in k),
            "v_permlane32_swap": sum(v for k, v in c.items() if "permlane32_swap" in k),
            "v_mov_b32_dpp": sum(v for k, v in c.items() if k.startswith("v_mov_b32_dpp")),
            "scratch_instructions": sum(v for k, v in c.items() if k.startswith("scratch_"))}


def registers(asm):
    """kernel symbol -> (vgpr_count incl. AGPRs, agpr_count, scratch bytes) from the code object metadata"""
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.vgpr_count:\s+(\d+)", asm, re.S):
        blk = m.group(2)
        ag = re.search(r"\.agpr_count:\s+(\d+)", blk); sc = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        out[m.group(1)] = {"vgpr_plus_agpr": int(m.group(3)), "agpr": int(ag.group(1)) if ag else 0, "scratch_bytes": int(sc.group(1)) if sc else 0}
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    bodies, regs = {}, {}
    with tempfile.TemporaryDirectory() as td:
        for tu in ("ccal_kernels_fused", "ccal_kernels_gram2", "ccal_kernels_schurq"):
            s = os.path.join(td, tu + ".s")
            # (-DCCAL_LEGACY_KERNELS: the separate elimination kernel k_schur1m - whose body is what the Gram kernels' fused tail runs -
            #  and the superseded Gram kernels exist in the second library only)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-S",
                                   "-DCCAL_LEGACY_KERNELS", "-DCCAL_DEV_SWITCHES", "--cuda-device-only", "-o", s, os.path.join(SRC, tu + ".hip")], stderr=subprocess.DEVNULL)
            asm = open(s).read()
            bodies.update(kernel_bodies(asm)); regs.update(registers(asm))
    out = {"source": "gfx950 ISA of ccal_kernels_fused.hip and ccal_kernels_gram2.hip (hipcc -O3 -ffp-contract=fast), counted by tools/count_flops.py",
           "fp64_vector_peak_tflops": 78.6, "kernels": {}}
    # model ids: 0 UCM 1 EUCM 2 KB4 3 OPENCV5; k_gram1w<MODEL, OF, LPF>, k_gram1v<...>, k_gram1<MODEL, OF>, k_schur1m<K>
    names = {0: "UCM", 1: "EUCM", 2: "KB4", 3: "OPENCV5"}
    want = {}
    for m, nm in names.items():
        for of in (0, 1):
            focal = "one-focal" if of else "two-focal"
            # the per-corner count does not depend on the lanes-per-frame instantiation: read it off the 12-lane one
            want[f"k_gram1w<{nm},{focal}>"] = f"k_gram1wILi{m}ELb{of}ELi12ELb0EE"
            want[f"k_gram1v<{nm},{focal}>"] = f"k_gram1vILi{m}ELb{of}ELi12ELb0ELb0EE"       # <MODEL, OF, LPF, GEN = false, ITER = false>
            want[f"k_gram1<{nm},{focal}>"] = f"k_gram1ILi{m}ELb{of}EE"
            want[f"k_gram2<{nm},{focal}>"] = f"k_gram2ILi{m}ELb{of}ELi12ELb0EE"         # <MODEL, OF, LPF, GEN = false>
            # KB4 / OPENCV5 run 10 frames per wavefront at 10 000 frames (6 lanes per frame): the instantiation the bench times
            want[f"k_gram1v<{nm},{focal},6 lanes>"] = f"k_gram1vILi{m}ELb{of}ELi6ELb0ELb0EE"
            want[f"k_gram2<{nm},{focal},6 lanes>"] = f"k_gram2ILi{m}ELb{of}ELi6ELb0EE"
    for name, key in want.items():
        hits = [k for k in bodies if key in k]
        if not hits:
            continue
        body = bodies[hits[0]]
        def n_fma(a, b):
            return sum(1 for x in body[a:b + 1] if "fma_f64" in x or "fmac_f64" in x)
        ls = [(a, b) for a, b in loops(body) if n_fma(a, b) >= 100]     # loops that hold a whole corner evaluation
        if not ls:
            continue
        a, b = min(ls, key=lambda t: t[1] - t[0])                      # the innermost of them: one corner per lane and trip
        loop = summarize(ops(body[a:b + 1]))
        whole = summarize(ops(body))
        mf = ops(body[a:b + 1]).get("v_mfma_f64_16x16x4_f64", 0)
        out["kernels"][name] = {"per_corner": loop, "outside_corner_loop_per_wavefront": {k: whole[k] - loop[k] for k in whole},
                                "mfma_f64_16x16x4_in_loop": mf, "registers": regs.get(hits[0])}
    for K in (5, 6, 7, 8, 9):
        hits = [k for k in bodies if f"k_schur1mILi{K}EE" in k]
        if hits:
            w = summarize(ops(bodies[hits[0]]))
            w["per_frame_flops_16_lanes"] = 16 * w["flops"]
            out["kernels"][f"k_schur1m<K={K}>"] = {"per_lane_whole_kernel": w}
    # k_schurq<PE>: straight-line code, one wavefront = 16 frame slots of a two-camera rig (4 lanes per slot)
    for PE in (4, 5, 6):
        hits = [k for k in bodies if f"k_schurqILi{PE}EE" in k]
        if hits:
            w = summarize(ops(bodies[hits[0]]))
            w["per_slot_flops_4_lanes"] = 4 * w["flops"]
            out["kernels"][f"k_schurq<PE={PE}>"] = {"per_lane_whole_kernel": w, "registers": regs.get(hits[0])}
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "flops.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out["kernels"].items():
        print(k, json.dumps(v)[:260])


if __name__ == "__main__":
    main()
