#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: Gbp of target aligned per second (whole job),
workload = BASELINE.json configs[1]: synthetic 50 Mbp target vs 50 Mbp query, default 12-of-19
seed with one transition, --nogapped (the HSP kernel path), both strands.

A step = one complete pass of the hot path over the batch, inputs already resident in HBM:
position-table build from the resident target (B1), then seed-hit search + X-drop extension of the
+ strand and of the - strand of this rank's query (B2), HSPs delivered to the host in the
reference's order.  N>1 (weak scaling): every rank owns its own 50 Mbp query (queries shard with no
data-path collective); per step rank 0 rebuilds the table and broadcasts it over RCCL/xGMI.

One JSON line on rank 0 (see the driver contract); extra objects: roofline, cpu_baseline.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


class _DevMem:
    """raw device pointer -> torch tensor view (zero copy) via __cuda_array_interface__"""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this workload
    (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, profiles/*pmc_fetch_write.csv).  Units and the
    gfx950 correction follow MI355X_MICROARCH.md: counters are in KiB; FETCH_SIZE under-reports wide
    reads by 2x (uncalibrated for 16-byte gathers: reported as measured x2 = upper bound)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_fetch_write.csv")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])   # r01_v10 after r01_v9
    if not files:
        return None, None
    for line in open(files[-1]).read().split("\n")[1:]:
        f = line.split(",")
        if len(f) >= 4 and f[0] == kernel:
            return (2.0 * float(f[2]) + float(f[3])) * 1024.0, os.path.basename(files[-1])
    return None, None


def cpu_baseline(t, q, qlen_bench, sample_bp, gapped=False):
    """The pristine reference (oracle/_ref/lastz_stats, built from /root/reference in the build
    container and shipped with the snapshot) on a bounded sample of the same workload, 1 core
    (lastz is single-threaded).  Falls back to the oracle port if the binary is absent."""
    from lastz_amd import seqio
    ts, qs = t[:sample_bp], q[:sample_bp]
    ref = os.path.join(ROOT, "oracle", "_ref", "lastz_stats")
    bp2 = float(len(ts)) * float(len(qs)) * 2.0
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory() as d:
            tf, qf, st = os.path.join(d, "t.fa"), os.path.join(d, "q.fa"), os.path.join(d, "st.txt")
            seqio.write_fasta(tf, [("target", ts)]); seqio.write_fasta(qf, [("query", qs)])
            t0 = time.time()
            # with --gapped the same run also yields the DP stage's clock and cell count (the table and search
            # clocks do not depend on what follows them)
            p = subprocess.run([ref, tf, qf, "--ydrop=9430" if gapped else "--nogapped", "--stats=" + st],
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            dp_cells = None
            if gapped and os.path.exists(st):
                for line in open(st):
                    if "DP cells visited" in line:
                        dp_cells = int(line.split(":")[1].replace(",", ""))
            wall = time.time() - t0
            clocks = {}
            for line in p.stderr.split("\n"):
                if ":" in line:
                    k, v = line.rsplit(":", 1)
                    try:
                        clocks[k.strip()] = float(v.split()[0])
                    except (ValueError, IndexError):
                        pass
        sec = clocks.get("seed position table", 0.0) + clocks.get("seed hit search", 0.0)
        if sec <= 0:
            sec = wall
        kind = "reference"
        detail = {"seed_position_table_s": clocks.get("seed position table"),
                  "seed_hit_search_s": clocks.get("seed hit search"), "process_wall_s": round(wall, 3)}
        if gapped and dp_cells and clocks.get("gapped extension"):
            detail["gapped"] = {"dp_cells": dp_cells, "gapped_extension_s": clocks["gapped extension"],
                                "gcups": dp_cells / clocks["gapped extension"] / 1e9, "cores": 1}
    else:
        from oracle import lzo
        _, masked = lzo.hoxd70_scoring()
        t0 = time.time()
        tab = lzo.Table(ts, lzo.seed())
        for qq in (qs, seqio.revcomp(qs)):
            lzo.seed_hit_search(tab, qq, masked)
        sec = time.time() - t0
        kind, detail = "port", {}
    rate_bp2 = bp2 / sec
    # seed work is proportional to Tlen*Qlen: express the CPU rate in the metric's unit AT THE BENCH
    # WORKLOAD's query size (Gbp of target per second against a qlen_bench query, both strands)
    value = rate_bp2 / (2.0 * qlen_bench) / 1e9
    return {"value": value, "unit": "Gbp/s", "cores": 1, "kind": kind,
            "sample": f"first {len(ts)} bp of target x first {len(qs)} bp of query, both strands, "
                      f"{'--ydrop=9430 run, seed-stage clocks' if gapped else '--nogapped'} ({sec:.2f} s of table+search CPU time); "
                      f"scaled by Tlen*Qlen to the bench query size",
            "bp2_per_s": rate_bp2, **detail}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tlen", type=int, default=50_000_000)
    ap.add_argument("--qlen", type=int, default=50_000_000)
    ap.add_argument("--cpu-sample", type=int, default=5_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gapped", action="store_true",
                    help="also time BASELINE.json configs[2]: the same pair with the gapped stage (--ydrop=9430)")
    a = ap.parse_args()

    import torch                                   # before liblzgpu.so: one HIP runtime per process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  LZ_BENCH_BACKEND=gloo exists only to exercise this code path with two
        # ranks on a one-GPU box (RCCL refuses two ranks on the same device); it is not a measured mode.
        backend = os.environ.get("LZ_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        dist = None

    from lastz_amd import lzgpu, seqio
    lib = lzgpu.Lib()
    lib.init(local)

    # ---- synthetic inputs (SURVEY 8d generator); the target is identical on every rank, each rank
    # gets its own query unit
    target, q0 = seqio.synth_pair(a.tlen, a.qlen, seed=1000)
    if rank == 0:
        query = q0
    else:
        rng_target = target
        _, query = None, None
        # homologous to the SAME target: regenerate the query blocks with a rank-specific stream
        rng = np.random.default_rng(2000 + rank)
        parts, have = [], 0
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        while have < a.qlen:
            blen = min(int(rng.integers(2000, 20001)), a.qlen - have, a.tlen)
            if rng.random() < 0.5 and blen >= 64:
                s = int(rng.integers(0, a.tlen - blen + 1))
                blk = seqio._mutate(rng, rng_target[s:s + blen], 0.12, 0.01)
                if rng.random() < 0.5:
                    blk = seqio.revcomp(blk)
            else:
                blk = acgt[rng.integers(0, 4, blen)]
            parts.append(blk); have += len(blk)
        query = np.concatenate(parts)[:a.qlen].copy()

    ctb = np.full(256, -1, dtype=np.int8)
    for i, ch in enumerate(b"ACGT"):
        ctb[ch] = i
    # lastz default scoring (HOXD70, lower case / N / X penalised in the HSP stage); built here
    # without the oracle: src/dna_utilities.c:137-148,215-300,497-552
    sub = np.full((256, 256), -100, dtype=np.int32)
    sub[0, :] = -107374182; sub[:, 0] = -107374182
    for ch in b"Xx":
        sub[ch, :] = -1000; sub[:, ch] = -1000
    hox = [[91, -114, -31, -123], [-114, 100, -125, -31], [-31, -125, 100, -114], [-123, -31, -114, 91]]
    for i, r in enumerate(b"ACGT"):
        for j, c in enumerate(b"ACGT"):
            for rr in (r, r + 32):
                for cc in (c, c + 32):
                    sub[rr, cc] = hox[i][j]
    masked = sub.copy()
    for ch in b"acgtNnX":
        masked[ch, 1:] = -1000
    for ch in b"acgtNnX":
        masked[1:, ch] = -1000

    sd = lib.seed("1110100110010101111", 1)
    if rank == 0:
        lib.table_prepare(target, sd, ctb)
    if world > 1:
        geom = [lib.table_geom() if rank == 0 else None]
        gb = [bytes(geom[0]) if rank == 0 else None]
        dist.broadcast_object_list(gb, src=0)
        if rank != 0:
            g = lzgpu.TableGeom.from_buffer_copy(gb[0])
            lib.table_adopt(g)
    lib.query_upload(0, query)
    lib.query_upload(1, seqio.revcomp(query))

    def bcast_table():
        from lastz_amd import shard
        for ptr, nbytes in lib.table_buffers():
            if nbytes == 0:
                continue
            try:
                tt = torch.as_tensor(_DevMem(ptr, nbytes), device=torch.device("cuda", local))
                shard.broadcast_buffers(dist, [tt], src=0)            # RCCL over xGMI, zero copy
            except Exception:
                stage = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", local))
                if rank == 0:
                    lib.device_copy(stage.data_ptr(), ptr, nbytes)
                dist.broadcast(stage, src=0)
                torch.cuda.synchronize()
                if rank != 0:
                    lib.device_copy(ptr, stage.data_ptr(), nbytes)
        torch.cuda.synchronize()

    n_hsps = [0]

    def step():
        if rank == 0:
            lib.table_rebuild()
        if world > 1:
            bcast_table()
            if rank != 0:
                lib.table_commit()
        n = 0
        for slot in (0, 1):
            n += len(lib.seed_hit_search(masked, slot=slot))
        n_hsps[0] = n

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        bcast_table()
        if rank != 0:
            lib.table_commit()
    for _ in range(a.warmup):
        step()
    lib.profile_enable(True); lib.profile_reset(); lib.counters_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    lib.profile_enable(False)

    gapped = None
    if a.gapped:
        # configs[2]: HSPs of each strand -> reduce_to_points -> gapped_extend (Y-drop DP), --ydrop=9430
        seed_prof, seed_cnt = lib.profile(), lib.counters()      # the main line's numbers, before this leg
        hsps = [lib.seed_hit_search(masked, slot=slot) for slot in (0, 1)]
        segs = []
        for rev, h in enumerate(hsps):
            sg = np.zeros(len(h), dtype=lzgpu.SEG_DTYPE)
            sg["pos1"] = h["pos1"] - h["length"]; sg["pos2"] = h["pos2"] - h["length"]
            sg["length"] = h["length"]; sg["s"] = h["score"]; sg["id"] = rev
            segs.append(sg)
        for slot in (0, 1):
            lib.gapped_extend(sub, segs[slot], slot=slot, ydrop=9430)              # warm-up (allocations)
        lib.profile_enable(True); lib.profile_reset(); lib.counters_reset()
        fence()
        g0 = time.perf_counter()
        nblocks = 0
        for slot in (0, 1):
            al, _ = lib.gapped_extend(sub, segs[slot], slot=slot, ydrop=9430)
            nblocks += len(al)
        fence()
        gdt = time.perf_counter() - g0
        gp, gc = lib.profile(), lib.counters()
        lib.profile_enable(False)
        kms = gp.get("k_ydrop", {"ms": 0.0, "launches": 0})
        gapped = {"workload": "BASELINE.json configs[2]: same pair, gapped stage, --ydrop=9430, both strands",
                  "wall_s": gdt, "anchors": int(len(segs[0]) + len(segs[1])), "alignments": nblocks,
                  "anchors_extended": gc["anchors_extended"], "dp_launched": gc["gapped_extensions"],
                  "dp_cells_reference": gc["dp_cells"], "gcups_wall": gc["dp_cells"] / gdt / 1e9,
                  "k_ydrop_ms": kms["ms"], "k_ydrop_launches": kms["launches"],
                  # algorithmic bytes of the DP, SURVEY 8(d): 1 traceback byte per visited cell
                  "roofline": {"bound": "hbm", "kernel": "k_ydrop", "achieved": (gc["dp_cells"] / (kms["ms"] * 1e-3) / 1e9) if kms["ms"] else None,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": (gc["dp_cells"] / (kms["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if kms["ms"] else None,
                               "traffic": None}}
        # restore the seed-stage profile for the main line
        class _P:  # noqa: E306
            pass
        lib.profile = lambda: seed_prof
        lib.counters = lambda: seed_cnt

    if rank == 0:
        prof = lib.profile()
        cnt = lib.counters()
        K = max(a.steps, 1)
        W, Hh, E, X = (cnt[k] / K for k in ("words", "raw_hits", "extensions", "bp_extended"))
        # algorithmic bytes per step (both strands), SURVEY.md 8(d): B_seed = W*(1+4V) + 8H + 4E + X
        V = sd.num_probes
        # split over the kernels that do each part: the table probes and chain links (count, fill), the
        # bases the X-drop scans touch (phase A = k_probe_part, which scans every hit and partitions the records),
        # the diagEnd read / write per hit / extension (phase B = k_settle, which settles hits from the summaries)
        alg = {"k_count_hits": W * (1 + 4 * V), "k_fill_hits": 4 * Hh, "k_probe_part": X, "k_settle": 4 * Hh + 4 * E}
        b_seed = W * (1 + 4 * V) + 8 * Hh + 4 * E + X
        kern_ms = {k: v["ms"] / K for k, v in prof.items()}
        dom = max((k for k in kern_ms if k in alg), key=lambda k: kern_ms[k], default=None)
        roof = None
        if dom:
            launches = prof[dom]["launches"] / K
            avg_ms = prof[dom]["ms"] / max(prof[dom]["launches"], 1)
            ach = alg[dom] / launches / (avg_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(dom)
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": alg[dom] / launches, "avg_launch_ms": avg_ms,
                    "launches_per_step": launches,
                    # whole seed stage against the same roofline, on WALL time (phase B of a chunk overlaps the
                    # next chunk's fill/probe/sort on a second stream, so per-kernel event times add up to more)
                    "stage": {"b_seed_bytes_per_step": b_seed, "wall_ms_per_step": dt / K * 1e3,
                              "sum_kernel_ms_per_step": sum(kern_ms.values()),
                              "frac_of_hbm_peak": b_seed / (dt / K) / 1e9 / HBM_PEAK_GBS}}
        ms_per_step = dt / K * 1e3
        value = world * (a.tlen / 1e9) / (dt / K)
        out = {"metric": "Gbp-of-target aligned/sec (whole job, --nogapped HSP path, both strands)",
               "value": value, "unit": "Gbp/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "s32", "data": "synthetic",
               "config": {"workload": "BASELINE.json configs[1]: synthetic %d bp target vs %d bp query per GPU, "
                                      "12-of-19 seed + 1 transition, --nogapped, both strands" % (a.tlen, a.qlen),
                          "tlen": a.tlen, "qlen_per_gpu": a.qlen, "parallelism": "query units sharded over %d GPU(s), "
                          "position table built on rank 0%s" % (world, " and RCCL-broadcast each step" if world > 1 else "")},
               "bp2_per_s": world * float(a.tlen) * float(a.qlen) * 2.0 / (dt / K),
               "hsps_rank0": n_hsps[0],
               "counters_per_step": {"words": W, "raw_hits": Hh, "extensions": E, "bp_extended": X},
               "kernel_ms_per_step": kern_ms, "roofline": roof}
        if gapped is not None:
            out["gapped"] = gapped
        if world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline(target, q0, a.qlen, min(a.cpu_sample, a.tlen, a.qlen), gapped=a.gapped)
            if gapped is not None and "gapped" in cb:
                gapped["cpu_baseline"] = cb.pop("gapped")
                gapped["speedup_vs_cpu_1core"] = gapped["gcups_wall"] / gapped["cpu_baseline"]["gcups"]
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_1core"] = value / cb["value"] if cb["value"] else None
        print(json.dumps(out))
    lib.shutdown()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
