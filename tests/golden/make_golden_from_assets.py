"""Derive golden board-frames from the reference's own artifacts (run in the build container only).

Input  (never copied into this repo): /root/reference/assets/2048_{drul,random}_actions.svg --
animations written by the reference's notebooks/explore_naive_strategies.ipynb via
``run_actions_batch(INIT_SEED=0, 4, act_fn)`` -> ``pgx.save_svg_animation`` (frame k = the 4 boards
AFTER step k; the init state is not a frame, reference src/runs/run_actions_batch.py:47-55).

Output (data only): tests/golden/svg_{drul,random}_seed0_b4.npy, uint8 [frames, 4 envs, 16 cells]
of log2(tile) (0 = empty), row-major.  Parse recipe: SURVEY.md Appendix A.5.

Also writes readme_histograms.json: the 1000-episode max-tile percentages printed in the reference
README figures (assets/{random,drul}_strategy_statistics.png; protocol
run/viz_naive_strategies.py:158-171: seed 42, 10 batches x 100 envs, batch seed 42 + 100*i).
"""
import json
import os
import re

import numpy as np

ASSETS = "/root/reference/assets"
HERE = os.path.dirname(os.path.abspath(__file__))

_FRAME = re.compile(r'<g class="frame" id="_fr([0-9a-f]+)"')
_BOARD = re.compile(r'<g transform="translate\(([0-9.]+),([0-9.]+)\)">(.*?)</g>', re.S)
_CELL = re.compile(r'<rect [^>]*?x="(\d+)" y="(\d+)" />(?:<text [^>]*>(\d+)</text>)?')
_ORIGINS = {(25.0, 25.0): 0, (275.0, 25.0): 1, (25.0, 275.0): 2, (275.0, 275.0): 3}


def parse(path: str) -> np.ndarray:
    text = open(path).read()
    starts = [(m.start(), int(m.group(1), 16)) for m in _FRAME.finditer(text)]
    assert [i for _, i in starts] == list(range(len(starts))), "frames out of order"
    frames = np.zeros((len(starts), 4, 16), dtype=np.uint8)
    for n, (pos, idx) in enumerate(starts):
        end = starts[n + 1][0] if n + 1 < len(starts) else len(text)
        seen = set()
        for bm in _BOARD.finditer(text[pos:end]):
            env = _ORIGINS[(float(bm.group(1)), float(bm.group(2)))]
            cells = _CELL.findall(bm.group(3))
            assert len(cells) == 16
            for x, y, val in cells:
                c, r = (int(x) - 2) // 50, (int(y) - 2) // 50
                if val:
                    v = int(val)
                    assert v & (v - 1) == 0
                    frames[idx, env, 4 * r + c] = v.bit_length() - 1
            seen.add(env)
        assert seen == {0, 1, 2, 3}
    return frames


if __name__ == "__main__":
    for name in ("drul", "random"):
        fr = parse(os.path.join(ASSETS, f"2048_{name}_actions.svg"))
        np.save(os.path.join(HERE, f"svg_{name}_seed0_b4.npy"), fr)
        print(name, fr.shape)
    hist = {
        "protocol": {"seed": 42, "batches": 10, "batch_size": 100, "batch_seed": "42 + 100*i",
                     "rng_mode": "partitionable", "source": "reference README.md:85-95"},
        "random_percent": {"16": 0.7, "32": 5.8, "64": 34.9, "128": 50.9, "256": 7.7},
        "drul_percent": {"32": 1.6, "64": 10.0, "128": 39.8, "256": 45.8, "512": 2.8},
    }
    json.dump(hist, open(os.path.join(HERE, "readme_histograms.json"), "w"), indent=1)
