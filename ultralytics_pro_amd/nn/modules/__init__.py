"""Operator library of the MI355X-native build: the class names a reference model YAML refers to
(ultralytics/nn/modules/__init__.py), each dispatching into libupa_hip.so."""

from .block import C2f, C3, DFL, MHSA, SPPF, BoT3, Bottleneck, BottleneckTransformer
from .conv import Concat, Conv, autopad
from .head import Detect

__all__ = ("Conv", "Concat", "autopad", "C2f", "C3", "DFL", "SPPF", "Bottleneck", "MHSA", "BottleneckTransformer",
           "BoT3", "Detect")
