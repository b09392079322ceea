"""Module API of the hot path (reference: src/model/*.py): same class names, constructor and
forward signatures and state_dict keys; every forward runs on the HIP kernels of libcst_hip.so."""
from .classifier import TextCNN
from .discriminator import RelGAN_D
from .match import Matcher
from .mlm import MLM
from .rnn import DenoiseLSTM

__all__ = ["DenoiseLSTM", "MLM", "Matcher", "TextCNN", "RelGAN_D"]
