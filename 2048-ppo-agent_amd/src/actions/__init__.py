from .act_drul import act_drul
from .act_randomly import act_randomly
