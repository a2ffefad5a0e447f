#!/usr/bin/env python3
"""10 Hz power / clock trace of a command (profiles/rNN_power*.csv): tools/power_trace.py <out.csv> <command...>

Reads the amdgpu hwmon / sysfs nodes of every GPU card directly (power1_average | power1_input in uW, freq1_input = sclk in Hz,
freq2_input = mclk, temp*_input, gpu_busy_percent) -- the rocm-smi CLI takes ~0.3 s per call, too slow for a 10 Hz trace; when no node is
readable it falls back to one `rocm-smi --json` sample per period.  Columns: seconds since start, then per card power_w, sclk_mhz,
mclk_mhz, busy_pct, temp_c.  The command's stdout / stderr pass through; exit code = the command's."""
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time


def cards():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if not hw or not os.path.exists(os.path.join(dev, "gpu_busy_percent")):
            continue
        out.append((os.path.basename(os.path.dirname(dev)), dev, hw[0]))
    return out


def rd(path, scale):
    try:
        return f"{float(open(path).read().split()[0]) * scale:.1f}"
    except Exception:
        return ""


def sample_sysfs(cs):
    row = []
    for _, dev, hw in cs:
        p = rd(os.path.join(hw, "power1_average"), 1e-6) or rd(os.path.join(hw, "power1_input"), 1e-6)
        row += [p, rd(os.path.join(hw, "freq1_input"), 1e-6), rd(os.path.join(hw, "freq2_input"), 1e-6), rd(os.path.join(dev, "gpu_busy_percent"), 1.0),
                rd(os.path.join(hw, "temp2_input"), 1e-3) or rd(os.path.join(hw, "temp1_input"), 1e-3)]
    return row


def sample_smi():
    try:
        d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse", "--showtemp", "--json"], capture_output=True, text=True, timeout=5).stdout)
    except Exception:
        return []
    row = []
    for card in sorted(d):
        def num(pat):
            for k, v in d[card].items():
                if re.search(pat, k, re.I):
                    m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
                    if m:
                        return m.group(0)
            return ""
        row += [num(r"power.*\(W\)"), num(r"sclk"), num(r"mclk"), num(r"GPU use"), num(r"Temperature.*(junction|hotspot|edge)")]
    return row


def hip_card():
    """sysfs card name of HIP device 0 (the node may show other tenants' GPUs too): PCI bus id from torch -> /sys/bus/pci/devices/*/drm/cardN"""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; p = torch.cuda.get_device_properties(0); "
                            "print('%04x:%02x:%02x' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, p.pci_device_id))"], capture_output=True, text=True, timeout=300)
        bdf = r.stdout.strip().splitlines()[-1]
        for d in glob.glob(f"/sys/bus/pci/devices/{bdf}.*/drm/card[0-9]*"):
            return os.path.basename(d)
    except Exception:
        pass
    return None


def card_of_device(index=0):
    """sysfs (card name, device dir, hwmon dir) of torch's HIP device `index` IN THIS PROCESS (PCI bus id -> /sys/bus/pci/devices/*/drm/cardN),
    or None when the nodes are not visible"""
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        for d in glob.glob(f"/sys/bus/pci/devices/{bdf}.*/drm/card[0-9]*"):
            name = os.path.basename(d)
            for c in cards():
                if c[0] == name:
                    return c
    except Exception:
        pass
    return None


class HwmonSampler:
    """background sampler of ONE GPU's socket power and shader clock (amdgpu hwmon: power1_average | power1_input, freq1_input) for the
    duration of a `with` block: bench.py brackets its headline pass with it, so that the roofline fraction of a power-capped kernel comes
    with the clock it was measured at.  .summary() -> dict (None values when the nodes are unreadable)"""

    def __init__(self, device_index=0, period_s=0.05):
        self.card, self.period, self.rows = card_of_device(device_index), period_s, []
        self._stop, self._th = threading.Event(), None

    def _read(self):
        _, dev, hw = self.card
        def num(path, scale):
            try:
                return float(open(path).read().split()[0]) * scale
            except Exception:
                return None
        p = num(os.path.join(hw, "power1_average"), 1e-6)
        if p is None:
            p = num(os.path.join(hw, "power1_input"), 1e-6)
        return p, num(os.path.join(hw, "freq1_input"), 1e-6), num(os.path.join(hw, "freq2_input"), 1e-6)

    def __enter__(self):
        self.rows = []
        if self.card is not None:
            self._stop.clear()

            def loop():
                while not self._stop.is_set():
                    self.rows.append(self._read())
                    self._stop.wait(self.period)
            self._th = threading.Thread(target=loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=2)
        return False

    def summary(self):
        def stat(i):
            v = [r[i] for r in self.rows if r[i] is not None and r[i] > 0]
            return (sum(v) / len(v), min(v), max(v)) if v else (None, None, None)
        (p, pmin, pmax), (f, fmin, fmax), (m, _, _) = stat(0), stat(1), stat(2)
        cap = None
        if self.card is not None:
            try:
                cap = float(open(os.path.join(self.card[2], "power1_cap")).read().split()[0]) * 1e-6
            except Exception:
                pass
        return {"samples": len(self.rows), "period_s": self.period, "card": self.card[0] if self.card else None,
                "socket_power_w": p, "socket_power_w_min_max": [pmin, pmax], "power_cap_w": cap,
                "sclk_mhz": f, "sclk_mhz_min_max": [fmin, fmax], "mclk_mhz": m,
                "source": "amdgpu hwmon (power1_average|power1_input, freq1_input) of the GPU this process computes on" if self.card else
                          "amdgpu hwmon nodes of this GPU are not readable from the process"}


_THROTTLE_CHILD = r"""
import json, sys
sys.path.insert(0, "/opt/rocm/share/amd_smi")
import amdsmi
amdsmi.amdsmi_init()
want = sys.argv[1].lower()
out = None
for h in amdsmi.amdsmi_get_processor_handles():
    try:
        bdf = amdsmi.amdsmi_get_gpu_device_bdf(h).lower()
    except Exception:
        continue
    if want and not bdf.startswith(want):
        continue
    v = amdsmi.amdsmi_get_violation_status(h)
    keep = ("acc_counter", "acc_prochot_thrm", "acc_ppt_pwr", "acc_socket_thrm", "acc_vr_thrm", "acc_hbm_thrm", "acc_gfx_clk_below_host_limit",
            "active_prochot_thrm", "active_ppt_pwr", "active_socket_thrm", "active_vr_thrm", "active_hbm_thrm")
    out = {k: (v[k] if isinstance(v.get(k), (int, float, bool)) else None) for k in keep}
    out["bdf"] = bdf
    break
print(json.dumps(out))
"""


def throttle_read(device_index=0):
    """the firmware's throttle-residency accumulators of ONE GPU (amdsmi violation status = the gpu_metrics accumulation counters: PPT / power,
    socket thermal, VR thermal, HBM thermal, PROCHOT), read in a child process so that the SMI library never shares an address space with HIP.
    -> dict or None.  Two reads around a pass give the share of the pass the firmware spent limiting the clock for each reason."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        r = subprocess.run([sys.executable, "-c", _THROTTLE_CHILD, bdf], capture_output=True, text=True, timeout=60)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        return None


def throttle_window(before, after):
    """residency of each limiter between two throttle_read() samples: accumulator delta / accumulation-counter delta"""
    if not before or not after or before.get("acc_counter") is None or after.get("acc_counter") is None:
        return {"available": False, "why_not": "amdsmi violation status (gpu_metrics throttle accumulators) not readable from this process"}
    dt = after["acc_counter"] - before["acc_counter"]
    out = {"available": True, "accumulation_ticks": dt, "source": "amdsmi_get_violation_status (amdgpu gpu_metrics throttler residency accumulators), read before / after the pass"}
    for k, name in (("acc_ppt_pwr", "ppt_power_limit"), ("acc_socket_thrm", "socket_thermal"), ("acc_vr_thrm", "vr_thermal"), ("acc_hbm_thrm", "hbm_thermal"),
                    ("acc_prochot_thrm", "prochot"), ("acc_gfx_clk_below_host_limit", "gfx_clk_below_host_limit")):
        a, b = before.get(k), after.get(k)
        out[name + "_residency"] = ((b - a) / dt) if (a is not None and b is not None and dt > 0) else None
    out["active_at_end"] = [k[7:] for k in ("active_ppt_pwr", "active_socket_thrm", "active_vr_thrm", "active_hbm_thrm", "active_prochot_thrm") if after.get(k)]
    return out


def main():
    out, cmd = sys.argv[1], sys.argv[2:]
    mine = hip_card()
    cs = cards()
    if mine is not None and any(c[0] == mine for c in cs):
        cs = [c for c in cs if c[0] == mine]        # only the GPU this process can see
    use_sysfs = bool(cs) and any(v for v in sample_sysfs(cs))
    names = [c[0] for c in cs] if use_sysfs else ["smi"]
    stop = threading.Event()

    def loop():
        with open(out, "w") as f:
            f.write("# source: " + ("amdgpu sysfs hwmon" if use_sysfs else "rocm-smi --json") + f"; HIP device 0 = {mine}\n")
            f.write("t_s," + ",".join(f"{n}_{c}" for n in names for c in ("power_w", "sclk_mhz", "mclk_mhz", "busy_pct", "temp_c")) + "\n")
            t0 = time.time()
            while not stop.is_set():
                row = sample_sysfs(cs) if use_sysfs else sample_smi()
                f.write(f"{time.time() - t0:.2f}," + ",".join(row) + "\n")
                f.flush()
                stop.wait(0.1)
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join(timeout=5)
    sys.exit(rc)


if __name__ == "__main__":
    main()
