from .batch_runner import BatchRunner, State
from .run_actions_batch import run_actions_batch
from .run_actions_max_tile import run_actions_max_tile
from .evaluate import evaluate_agent, evaluate_max_tile
from .svg_export import save_svg_animation, save_trajectory_npz, svg_animation
