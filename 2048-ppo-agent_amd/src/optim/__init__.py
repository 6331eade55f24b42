from .configure_optimizers import configure_bert_optimizers
from .lamb import Lamb
