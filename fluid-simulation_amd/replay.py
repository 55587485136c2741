"""Replay of a recorded interactive session.

The reference's only input path is GLUT: mouseClick stores the press position (kernel.cu:860-868),
every mouseMove turns the pixel delta since the last event into a force and splats it — always at
the window centre, the cursor-position call being commented out (kernel.cu:870-878) — and keys
'1'..'4' choose the dye colour, 'r' resets (kernel.cu:813-856).  One frame = one simulationStep
(display(), kernel.cu:786-801).  ``replay`` applies a list of such events to any object with the
FluidSim interface, converting exactly like the reference:

    force = (x - mX, -(y - mY))                        kernel.cu:872-873 (y flipped)
    addForce(height / 2, width / 2, force)             kernel.cu:875  (integer division)
    fx = x / width, fy = y / height                    kernel.cu:689-690 (float division)
"""
from __future__ import annotations

import json
from typing import Iterable

import numpy as np

from .splats import COLOR_PRESETS

F32 = np.float32


def parse_events(text: str):
    """JSON list of events, each {"frame": int, "type": "down"|"move"|"key", ...}:
    down/move carry pixel "x", "y"; key carries "key" ('1'..'4', 'r')."""
    ev = json.loads(text)
    return sorted(ev, key=lambda e: e["frame"])


def replay(sim, events: Iterable[dict], frames: int, width: int = 512, height: int = 512):
    """Run ``frames`` frames; events of frame f are applied before that frame's step, in order."""
    events = sorted(events, key=lambda e: e["frame"])
    color = COLOR_PRESETS[1]  # densityColor starts white, kernel.cu:507
    mx = my = 0               # mX, mY (kernel.cu:858) are zero-initialised globals
    i = 0
    for f in range(frames):
        while i < len(events) and events[i]["frame"] <= f:
            e = events[i]
            i += 1
            if e["type"] == "down":                      # mouseClick, left button down
                mx, my = int(e["x"]), int(e["y"])
            elif e["type"] == "move":                    # mouseMove
                x, y = int(e["x"]), int(e["y"])
                fx, fy = F32(x - mx), -F32(y - my)
                px, py = height // 2, width // 2         # addForce(height / 2, width / 2, ...)
                sim.add_source(float(F32(px) / F32(width)), float(F32(py) / F32(height)),
                               float(fx), float(fy), color)
                mx, my = x, y
            elif e["type"] == "key":
                k = str(e["key"])
                if k in "1234":
                    color = COLOR_PRESETS[int(k)]
                elif k == "r":
                    sim.reset()
            else:
                raise ValueError(f"unknown event type {e['type']!r}")
        sim.step(1)
