#!/usr/bin/env python3
"""Sample the GPU's shader clock, memory clock and power while a command runs.

  python scripts/clock_log.py [--period 0.02] [--out FILE.json] -- <command ...>

Sources, in order of preference: the amdgpu sysfs nodes of card 0 (gpu_metrics is avoided: its
layout is versioned) -- pp_dpm_sclk / pp_dpm_mclk (the starred level), hwmon freq1_input
(current sclk in Hz), hwmon power1_average / power1_input (microwatts); `rocm-smi` once at the
start and once at the end as a cross-check.  Prints a JSON summary: the distribution of the
samples taken while the GPU was busy (gpu_busy_percent > 0 when that node exists)."""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time


def find_nodes():
    nodes = {}
    cards = [c for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
             if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
    # several cards in sysfs, one visible to HIP: CLOCK_LOG_CARD picks it (index into `cards`);
    # default: the card with the highest power reading right now is unlikely to be right either,
    # so main() samples EVERY card and reports the one whose power moved most
    pick = int(os.environ.get("CLOCK_LOG_CARD", "0"))
    nodes["cards"] = cards
    for card in cards[pick:pick + 1]:
        nodes["card"] = card
        nodes["sclk"] = os.path.join(card, "pp_dpm_sclk")
        nodes["mclk"] = os.path.join(card, "pp_dpm_mclk")
        nodes["busy"] = os.path.join(card, "gpu_busy_percent")
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for name in ("freq1_input", "power1_average", "power1_input", "temp1_input"):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    nodes[name] = p
        break
    return nodes


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def starred_mhz(text):
    if not text:
        return None
    for line in text.splitlines():
        if "*" in line:
            try:
                return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            except (IndexError, ValueError):
                return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--period", type=float, default=0.02)
    ap.add_argument("--out", default=None)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    nodes = find_nodes()
    if "CLOCK_LOG_CARD" not in os.environ and len(nodes.get("cards", [])) > 1:
        # probe: run nothing, look at every card for ~0.3 s while a tiny HIP job spins?  Simpler:
        # sample all cards during the command and keep the one with the largest power swing.
        all_nodes = []
        for i in range(len(nodes["cards"])):
            os.environ["CLOCK_LOG_CARD"] = str(i)
            all_nodes.append(find_nodes())
        del os.environ["CLOCK_LOG_CARD"]
    else:
        all_nodes = [nodes]
    samples = []
    stop = threading.Event()

    per_card = [[] for _ in all_nodes]

    def sample_one(nodes):
        s = {"t": time.time()}
        if "freq1_input" in nodes:
            v = read(nodes["freq1_input"])
            s["sclk_mhz"] = float(v) / 1e6 if v else None
        else:
            s["sclk_mhz"] = starred_mhz(read(nodes.get("sclk", "")))
        s["mclk_mhz"] = starred_mhz(read(nodes.get("mclk", "")))
        for k in ("power1_average", "power1_input"):
            if k in nodes:
                v = read(nodes[k])
                s["power_w"] = float(v) / 1e6 if v else None
                break
        if "busy" in nodes:
            v = read(nodes["busy"])
            s["busy"] = float(v) if v and v.strip() else None
        return s

    def sampler_all():
        while not stop.is_set():
            for i, nd in enumerate(all_nodes):
                per_card[i].append(sample_one(nd))
            time.sleep(a.period)

    def sampler():
        while not stop.is_set():
            s = {"t": time.time()}
            if "freq1_input" in nodes:
                v = read(nodes["freq1_input"])
                s["sclk_mhz"] = float(v) / 1e6 if v else None
            else:
                s["sclk_mhz"] = starred_mhz(read(nodes.get("sclk", "")))
            s["mclk_mhz"] = starred_mhz(read(nodes.get("mclk", "")))
            for k in ("power1_average", "power1_input"):
                if k in nodes:
                    v = read(nodes[k])
                    s["power_w"] = float(v) / 1e6 if v else None
                    break
            if "busy" in nodes:
                v = read(nodes["busy"])
                s["busy"] = float(v) if v and v.strip() else None
            samples.append(s)
            time.sleep(a.period)

    def smi():
        try:
            return subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showuse"], capture_output=True,
                                  text=True, timeout=20).stdout[-1500:]
        except Exception as e:  # noqa: BLE001
            return f"rocm-smi unavailable: {e}"

    before = smi()
    th = threading.Thread(target=sampler_all if len(all_nodes) > 1 else sampler, daemon=True)
    th.start()
    t0 = time.time()
    rc = subprocess.call(cmd)
    t1 = time.time()
    stop.set()
    th.join(timeout=1)
    after = smi()
    if len(all_nodes) > 1:
        def swing(lst):
            p = [x["power_w"] for x in lst if x.get("power_w") is not None]
            return (max(p) - min(p)) if p else 0.0
        best = max(range(len(all_nodes)), key=lambda i: swing(per_card[i]))
        samples = per_card[best]
        nodes = all_nodes[best]
        nodes["picked_card_index"] = best
        nodes["power_swing_w_per_card"] = [round(swing(c), 1) for c in per_card]

    def dist(key, busy_only=True):
        vals = [s[key] for s in samples if s.get(key) is not None and (not busy_only or (s.get("busy") or 0) > 0
                                                                     or "busy" not in nodes)]
        if not vals:
            return None
        vals.sort()
        n = len(vals)
        return {"n": n, "min": vals[0], "p10": vals[n // 10], "median": vals[n // 2], "p90": vals[(9 * n) // 10],
                "max": vals[-1], "mean": sum(vals) / n}

    out = {"cmd": cmd, "rc": rc, "wall_s": round(t1 - t0, 2), "nodes": {k: v for k, v in nodes.items()},
           "samples": len(samples), "sclk_mhz_busy": dist("sclk_mhz"), "sclk_mhz_all": dist("sclk_mhz", False),
           "mclk_mhz_busy": dist("mclk_mhz"), "power_w_busy": dist("power_w"), "power_w_all": dist("power_w", False),
           "rocm_smi_before": before, "rocm_smi_after": after}
    txt = json.dumps(out, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(txt)
    print(txt, file=sys.stderr)
    sys.exit(rc)


if __name__ == "__main__":
    main()
