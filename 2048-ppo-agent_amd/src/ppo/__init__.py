from .data_loader import create_ppo_dataloader
from .ppo_agent import MLPAgent, PPOAgent
from .ppo_trainer import PPOTrainer
from .rollout_buffer import RolloutBuffer
from .torch_action_wrapper import TorchActionFunction
