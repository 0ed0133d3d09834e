#!/usr/bin/env python
"""
Clocks, power and temperatures of a GPU, read from sysfs (no HIP call, no subprocess):

    python tools/device_state.py [ordinal]      -> one JSON line

Used by bench.py (before / after the timed region) and by the profiling scripts, so that every
measured number carries the state of the device it ran on: across the MI355X devices of the pool
the same binary ranged 0.193-0.24 ms for the headline launch.

What is read (whatever exists; missing files are skipped):
  pp_dpm_sclk / pp_dpm_mclk / pp_dpm_fclk       current level of each clock domain (the starred row)
  hwmon: power1_average|power1_input, power1_cap  (microwatts -> W)
         temp*_input with temp*_label             (millidegrees -> C: edge / junction / mem)
         freq*_input with freq*_label             (Hz -> MHz: sclk / mclk)
  gpu_busy_percent, mem_busy_percent
"""

from __future__ import annotations

import json
import re
import sys
import time
from pathlib import Path


def _cards():
    cards = []
    for p in sorted(Path("/sys/class/drm").glob("card[0-9]*")):
        if re.fullmatch(r"card\d+", p.name) and (p / "device" / "pp_dpm_sclk").exists():
            cards.append(p / "device")
    return cards


def _read(p: Path):
    try:
        return p.read_text().strip()
    except OSError:
        return None


def _current_level(text):
    """'0: 132Mhz\\n1: 2400Mhz *' -> 2400"""
    if not text:
        return None
    for ln in text.splitlines():
        if ln.rstrip().endswith("*"):
            m = re.search(r"(\d+)\s*[Mm][Hh]z", ln)
            return int(m.group(1)) if m else None
    return None


def find_card(ordinal: int = 0, pci: tuple | None = None):
    """sysfs device directory of a GPU: by PCI (domain, bus, device) when given -- HIP ordinals and
    DRM card numbers need not agree -- else the ordinal-th card that has clock tables."""
    cards = _cards()
    if not cards:
        return None
    if pci is not None:
        want = "%04x:%02x:%02x." % tuple(int(x) for x in pci)
        for dev in cards:
            try:
                if dev.resolve().name.startswith(want):
                    return dev
            except OSError:
                pass
    return cards[min(ordinal, len(cards) - 1)]


def sample(ordinal: int = 0, pci: tuple | None = None) -> dict:
    dev = find_card(ordinal, pci)
    if dev is None:
        return {"error": "no amdgpu card with pp_dpm_sclk under /sys/class/drm"}
    out: dict = {"t": round(time.time(), 3), "card": dev.parent.name}
    for dom in ("sclk", "mclk", "fclk", "socclk"):
        lvl = _current_level(_read(dev / f"pp_dpm_{dom}"))
        if lvl is not None:
            out[f"dpm_{dom}_mhz"] = lvl
    for name in ("gpu_busy_percent", "mem_busy_percent"):
        v = _read(dev / name)
        if v is not None and v.lstrip("-").isdigit():
            out[name] = int(v)
    for hw in sorted((dev / "hwmon").glob("hwmon*")) if (dev / "hwmon").exists() else []:
        for f in sorted(hw.glob("power1_*")):
            if f.name in ("power1_average", "power1_input", "power1_cap"):
                v = _read(f)
                if v and v.isdigit():
                    out[f.name + "_w"] = round(int(v) * 1e-6, 1)
        for f in sorted(hw.glob("temp*_input")):
            v, label = _read(f), _read(f.with_name(f.name.replace("_input", "_label")))
            if v and v.lstrip("-").isdigit():
                out[f"temp_{label or f.name[:-6]}_c"] = round(int(v) * 1e-3, 1)
        for f in sorted(hw.glob("freq*_input")):
            v, label = _read(f), _read(f.with_name(f.name.replace("_input", "_label")))
            if v and v.isdigit():
                out[f"freq_{label or f.name[:-6]}_mhz"] = round(int(v) * 1e-6)
    return out


class Sampler:
    """Background sampling while a measurement runs (an instantaneous reading taken before or after it
    sees an idle device: the clocks fall back within milliseconds).  `with Sampler(0) as sm: ...;
    sm.summary()` -> mean / min / max of every numeric reading."""

    def __init__(self, ordinal: int = 0, pci: tuple | None = None, period_s: float = 0.02) -> None:
        import threading

        self.ordinal, self.pci, self.period = ordinal, pci, period_s
        self.rows: list = []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self) -> None:
        while not self._stop.is_set():
            self.rows.append(sample(self.ordinal, self.pci))
            self._stop.wait(self.period)

    def __enter__(self) -> "Sampler":
        self._thread.start()
        return self

    def __exit__(self, *exc) -> None:
        self._stop.set()
        self._thread.join(timeout=2.0)

    def summary(self) -> dict:
        out: dict = {"samples": len(self.rows)}
        if not self.rows:
            return out
        out["card"] = self.rows[0].get("card")
        keys = sorted({k for r in self.rows for k, v in r.items() if isinstance(v, (int, float)) and k != "t"})
        for k in keys:
            vals = [r[k] for r in self.rows if isinstance(r.get(k), (int, float))]
            if vals:
                out[k] = {"mean": round(sum(vals) / len(vals), 1), "min": min(vals), "max": max(vals)}
        return out


if __name__ == "__main__":
    print(json.dumps(sample(int(sys.argv[1]) if len(sys.argv) > 1 else 0)))
