"""SVG state-animation export (reference run/viz_naive_strategies.py:113-120): the golden frames
derived from the reference's own assets are written out and read back with the parser that derived them."""
import importlib.util
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


def _golden_parser():
    spec = importlib.util.spec_from_file_location("make_golden_from_assets", os.path.join(G, "make_golden_from_assets.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.parse


@pytest.mark.parametrize("name", ["drul", "random"])
def test_golden_frames_round_trip_through_svg(name, tmp_path):
    from src.runs.svg_export import save_svg_animation

    frames = np.load(os.path.join(G, f"svg_{name}_seed0_b4.npy"))  # [T, 4, 16]
    path = str(tmp_path / f"{name}.svg")
    save_svg_animation(frames, path, frame_duration_seconds=0.5)
    back = _golden_parser()(path)
    assert back.shape == frames.shape and (back == frames).all()


def test_state_list_and_layout(tmp_path):
    """A list of State-like snapshots (one-hot observation only) and a single board; 4 boards sit on a 2 x 2 grid."""
    from types import SimpleNamespace

    from src.runs.svg_export import svg_animation

    rng = np.random.default_rng(0)
    boards = rng.integers(0, 12, size=(3, 4, 16)).astype(np.uint8)
    obs = (boards[..., None] == np.arange(31)).reshape(3, 4, 4, 4, 31)
    states = [SimpleNamespace(observation=o) for o in obs]
    text = svg_animation(states)
    for origin in ("translate(25.0,25.0)", "translate(275.0,25.0)", "translate(25.0,275.0)", "translate(275.0,275.0)"):
        assert text.count(origin) == 3
    p = tmp_path / "s.svg"
    p.write_text(text)
    assert (_golden_parser()(str(p)) == boards).all()
    one = svg_animation(boards[:, 0])  # [T, 16] -> one board per frame
    assert one.count('<g class="frame"') == 3 and 'width="250.0"' in one
