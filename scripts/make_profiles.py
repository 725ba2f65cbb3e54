#!/usr/bin/env python3
"""Turn the raw output of scripts/collect_profiles.sh (gpurun_out/final) into the tracked summaries under profiles/.
usage: python scripts/make_profiles.py gpurun_out/final r01"""
import csv
import datetime
import glob
import json
import os
import shutil
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


# 1. bench lines
json.dump(last_json_line(f"{src}/bench.json"), open(f"{dst}/{tag}_bench_line.json", "w"), indent=1)
json.dump(last_json_line(f"{src}/bench_under_rocprof.json"), open(f"{dst}/{tag}_bench_line_under_rocprof.json", "w"), indent=1)

if os.path.exists(f"{src}/bench_extras.json"):
    shutil.copy(f"{src}/bench_extras.json", f"{dst}/{tag}_bench_extras.json")

# 2. kernel stats of the bench run (library kernels only)
rows, seen = [], set()
for f in sorted(glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime, reverse=True)[:1]:  # newest run only
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(("th::", "void th::")) and r["Name"] not in seen:
            seen.add(r["Name"])
            rows.append(r)
with open(f"{dst}/{tag}_bench_kernel_stats.csv", "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 "
             "--no-cpu-baseline --no-single-track\n# MI355X.  Library kernels only (torch kernels of the synthetic-signal "
             "generator omitted); durations in ns.\n")
    if rows:
        cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
        fo.write(",".join(cols) + "\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            fo.write(",".join('"' + r[c] + '"' if "," in r[c] else r[c] for c in cols) + "\n")

# 3. PMC summaries
hdr = ("# rocprofv3 --kernel-trace --pmc <one counter group per pass> -- python3 {script} --reps 3   (scripts/pmc_stft.sh)\n"
       "# per-dispatch medians over the launches of each kernel; SQ_* in quad-cycles summed over waves, FETCH_SIZE/WRITE_SIZE in KB.\n"
       "# gfx950: FETCH_SIZE reports exactly 1/2 of streamed bytes for 4/8/16-byte-per-lane reads "
       "(profiles/r01_fetch_calibration.txt); WRITE_SIZE is exact.\n")
for name, script in (("stft", "scripts/bench_stft.py"), ("stftpk", "scripts/bench_stft.py --kernel 9"),
                     ("stftsweep", "scripts/bench_stft.py --kernel 11"), ("img", "scripts/bench_img.py"),
                     ("stft1024", "scripts/bench_stft.py --nfft 1024"), ("stft512_multi", "scripts/bench_stft.py --nfft 512"),
                     ("stft4096", "scripts/bench_stft.py --nfft 4096 --seconds 60"),
                     ("stftmel", "scripts/bench_stft.py --sr 44100 --tracks 32 --seconds 60 --mel 128"),
                     ("stft4096dyn", "scripts/bench_stft.py --sr 96000 --nfft 4096 --win 3840 --hop 960 --seconds 30"),
                     ("stft512mel", "scripts/bench_stft.py --sr 8000 --nfft 512 --win 320 --hop 80 --mel 0 --seconds 180"),
                     ("stft4096mel96", "scripts/bench_stft.py --sr 96000 --nfft 4096 --win 3840 --hop 960 --seconds 30 --mel 0"),
                     ("melbandrows96", "scripts/bench_stft.py --sr 96000 --nfft 4096 --win 3840 --hop 960 --seconds 30 --mel 0 --kernel 12"),
                     ("stft4096mel48", "scripts/bench_stft.py --sr 48000 --nfft 4096 --mel 0"),
                     ("melmfma48", "scripts/bench_stft.py --sr 48000 --nfft 4096 --mel 0 --kernel 12"),
                     ("stftmel48", "scripts/bench_stft.py --sr 48000 --win 1920 --hop 480 --mel 0"),
                     ("stftmel48_one_frame", "scripts/bench_stft.py --sr 48000 --win 1920 --hop 480 --mel 0 --kernel 13"),
                     ("subwave32768", "scripts/bench_stft.py --nfft 32768"),
                     ("blockmel16384", "scripts/bench_stft.py --nfft 16384 --mel 0"), ("blockmel8192", "scripts/bench_stft.py --nfft 8192 --mel 0"),
                     ("block16384", "scripts/bench_stft.py --nfft 16384"),
                     ("melsmall2048", "scripts/bench_stft.py --sr 16000 --nfft 2048 --win 1360 --hop 340 --seconds 90 --mel 0"),
                     ("melsmall512", "scripts/bench_stft.py --sr 16000 --nfft 512 --win 320 --hop 80 --seconds 90 --mel 0"),
                     ("mel4096hop120", "scripts/bench_stft.py --sr 48000 --nfft 4096 --win 1920 --hop 120 --mel 0")):
    p = f"{src}/pmc_{name}/summary.txt"
    if os.path.exists(p):
        open(f"{dst}/{tag}_{name}_pmc_summary.txt", "w").write(hdr.format(script=script) + open(p).read())

# 4. HBM traffic of the dominant kernel (bench.py reads this file)
p = f"{src}/pmc_stft/summary.txt"
if os.path.exists(p):
    kern, vals = None, {}
    for line in open(p):
        if not line.startswith(" "):
            kern = line.strip()
            continue
        if kern and "stft_wave_kernel" in kern:
            c, rest = line.split(None, 1)
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[c] = float(rest.split("median=")[1].split()[0])
                vals["kernel"] = kern
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        frames, bpf = 128 * 2813, 6148
        json.dump({"kernel": vals["kernel"].replace("void th::", ""),
                   "bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
                   "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
                   "correction": "gfx950 FETCH_SIZE x2 (calibrated, profiles/r01_fetch_calibration.txt); WRITE_SIZE exact",
                   "workload": "128 tracks x 30 s 48 kHz mono, n_fft=2048 hop=512 (scripts/bench_stft.py, same shapes as bench.py)",
                   "algorithmic_bytes_per_launch": frames * bpf, "source": f"profiles/{tag}_stft_pmc_summary.txt",
                   "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes (scripts/pmc_stft.sh)",
                   "date": datetime.date.today().isoformat(),
                   "commit": subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True,
                                            text=True).stdout.strip() or "?"},
                  open(f"{dst}/stft_hbm_traffic.json", "w"), indent=1)

# 5. plain-text measurement logs
for f in ("bench_stft.txt", "bench_img.txt", "bench_cfg3.txt", "bench_cfg4.txt", "phase_prof.txt", "ubench_lds_rate.txt",
          "ubench_valu_rate.txt", "ubench_valu_bank.txt", "ubench_copy_rate.txt", "ubench_stream_shapes.txt", "ubench_fused_img_shapes.txt",
          "ubench_stft_skeleton_sweep.txt", "ubench_mom_probe.txt", "bluestein_probe.txt", "fuzz_mel_moments.txt", "wave_times.txt", "power.txt",
          "bench_line_force_dist.json", "bench_line_launcher.json", "bench_line_rehearsal_2_ranks_one_gpu.json", "bench_line_rehearsal_4_ranks_one_gpu.json", "gputest.txt", "box.txt", "build_mode.txt"):
    if os.path.exists(f"{src}/{f}"):
        txt = "\n".join(l for l in open(f"{src}/{f}").read().splitlines() if "amdgpu.ids" not in l) + "\n"
        open(f"{dst}/{tag}_{f}", "w").write(txt)
print("\n".join(sorted(os.listdir(dst))))

# 6. the N > 1 launcher refusing a one-GPU box (round 5): bench.py --gpus 2 -> stderr + exit code, no record on stdout
if os.path.exists(f"{src}/gpus2.err"):
    keep = [l for l in open(f"{src}/gpus2.err").read().splitlines() if "bench.py" in l or l.startswith("exit code")]
    open(f"{dst}/{tag}_bench_gpus2_on_one_gpu.txt", "w").write("# python3 bench.py --gpus 2 --steps 2 on a one-GPU box: must refuse (stderr, exit code != 0, empty stdout)\n" + "\n".join(keep) + "\n")
