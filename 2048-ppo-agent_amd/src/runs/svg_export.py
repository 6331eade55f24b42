"""Trajectory -> animated SVG, the state-animation export of the reference's visualisation scripts
(``pgx.save_svg_animation(states, output_path, frame_duration_seconds=0.25)`` at run/viz_naive_strategies.py:113-120,
run/viz_ppo_agent.py; the files under the reference's assets/ are its output).

The document structure is the one SURVEY.md Appendix A.5 describes, so the parser that extracts the golden boards from
the reference's assets (tests/golden/make_golden_from_assets.py) reads these files too: one ``<svg>``, one
``<g class="frame" id="_fr{hex}">`` per step shown in turn by a CSS keyframe animation, inside a frame one
``<g transform="translate(x,y)">`` per board on a square grid of 250-pixel cells (board origin at +25,+25), inside a
board 16 ``<rect x="2+50c" y="2+50r">`` in row-major order, each followed by ``<text>VALUE</text>`` unless the cell is
empty.  Colours and fonts are this module's own (grey level by exponent).
"""
from __future__ import annotations

import math
from typing import Iterable, Sequence, Union

import numpy as np

CELL, PAD, BOARD_PX = 50, 25, 250


def _as_frames(states_or_frames) -> np.ndarray:
    """-> u8 [T, B, 16] of log2(tile).  Accepts that array (numpy / torch), a list of ``State`` snapshots (their
    ``board`` [B, 16], or ``observation`` [B, 4, 4, 31] one-hot), or an engine ``Trajectory`` (frames after each step)."""
    x = states_or_frames
    if hasattr(x, "boards") and hasattr(x, "final_boards"):  # engine Trajectory: state AFTER step k = board before k+1
        import torch

        x = torch.cat([x.boards[1:], x.final_boards[None]], dim=0)
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    if isinstance(x, np.ndarray):
        fr = x
    else:
        rows = []
        for s in x:
            b = getattr(s, "board", None)
            if b is None:
                b = np.asarray(s.observation).argmax(axis=-1).reshape(-1, 16)
            rows.append(np.asarray(b).reshape(-1, 16))
        fr = np.stack(rows)
    fr = np.asarray(fr)
    if fr.ndim == 2:
        fr = fr[:, None, :]
    if fr.ndim == 4:  # [T, B, 4, 4]
        fr = fr.reshape(fr.shape[0], fr.shape[1], 16)
    if fr.ndim != 3 or fr.shape[-1] != 16:
        raise ValueError(f"expected frames [T, B, 16], got {fr.shape}")
    return fr.astype(np.uint8)


def _cell(e: int, c: int, r: int) -> str:
    grey = max(36, 242 - 22 * e)
    fill = f"#{grey:02x}{grey:02x}{grey:02x}"
    out = (f'<rect fill="{fill}" height="46" rx="3px" ry="3px" stroke="black" stroke-width="0.5px" width="46" '
           f'x="{2 + CELL * c}" y="{2 + CELL * r}" />')
    if e:
        txt = str(1 << e)
        ink = "black" if grey > 128 else "#e1e1e1"
        tx = 2 + CELL * c + 23 - 5.4 * len(txt)
        out += (f'<text fill="{ink}" font-family="Courier" font-size="18px" font-weight="bold" x="{tx:.1f}" '
                f'y="{30.0 + CELL * r:.1f}">{txt}</text>')
    return out


def svg_animation(states_or_frames, frame_duration_seconds: float = 0.25) -> str:
    fr = _as_frames(states_or_frames)
    T, B, _ = fr.shape
    cols = max(1, math.ceil(math.sqrt(B)))
    rows = math.ceil(B / cols)
    W, H = BOARD_PX * cols, BOARD_PX * rows
    total = max(T, 1) * frame_duration_seconds
    pct = 100.0 / max(T, 1)
    css = [f".frame{{visibility:hidden; animation:{total}s linear _k infinite;}}",
           f"@keyframes _k{{0%,{pct}%{{visibility:visible}}{pct * 1.000001}%,100%{{visibility:hidden}}}}"]
    css += [f"#_fr{t:x}{{animation-delay:{t * frame_duration_seconds}s}}" for t in range(T)]
    parts = ['<?xml version="1.0" encoding="utf-8" ?>\n',
             f'<svg baseProfile="full" height="{float(H)}" version="1.1" width="{float(W)}" '
             'xmlns="http://www.w3.org/2000/svg" xmlns:ev="http://www.w3.org/2001/xml-events" '
             'xmlns:xlink="http://www.w3.org/1999/xlink"><defs><style type="text/css"><![CDATA[', "".join(css),
             "]]></style></defs>", f'<rect fill="white" height="{H}" width="{W}" x="0" y="0" />']
    for t in range(T):
        parts.append(f'<g class="frame" id="_fr{t:x}" transform="scale(1.0)">'
                     f'<rect fill="white" height="{H}" width="{W}" x="0" y="0" />')
        for b in range(B):
            gx, gy = (b % cols) * BOARD_PX, (b // cols) * BOARD_PX
            parts.append(f'<g transform="translate({float(gx + PAD)},{float(gy + PAD)})">')
            parts.extend(_cell(int(fr[t, b, 4 * r + c]), c, r) for r in range(4) for c in range(4))
            parts.append(f'</g><rect fill="none" height="{BOARD_PX}" stroke="gray" width="{BOARD_PX}" x="{gx}" y="{gy}" />')
        parts.append("</g>")
    parts.append("</svg>")
    return "".join(parts)


def save_svg_animation(states_or_frames, filename: str, *, frame_duration_seconds: float = 0.25) -> None:
    """Same call shape as ``pgx.save_svg_animation`` (run/viz_naive_strategies.py:113-120)."""
    with open(filename, "w") as f:
        f.write(svg_animation(states_or_frames, frame_duration_seconds))


def save_trajectory_npz(traj, filename: str) -> None:
    """Raw trajectory export (step-major arrays of an engine ``Trajectory``) for offline analysis."""
    to = lambda x: None if x is None else x.detach().cpu().numpy()
    arrays = dict(boards=to(traj.boards), actions=to(traj.actions), masks=to(traj.masks), terms=to(traj.terms),
                  rewards=to(traj.rewards), ep_len=to(traj.ep_len), final_boards=to(traj.final_boards),
                  init_boards=to(traj.init_boards))
    for k in ("log_probs", "values"):
        v = to(getattr(traj, k))
        if v is not None:
            arrays[k] = v
    np.savez_compressed(filename, **arrays)
