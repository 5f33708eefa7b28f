"""Predictor hot spots: the text-augmented encoder-decoder of
textreact/model.py with its attention and add+LayerNorm running as hand-written gfx950 kernels."""
