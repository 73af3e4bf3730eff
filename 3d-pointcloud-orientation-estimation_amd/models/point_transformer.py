"""models/point_transformer.py -- drop-in for the reference file of the same name (models/point_transformer.py:4-20).

The parameter containers are the reference's own modules in the same construction order (nn.Linear,
nn.TransformerEncoderLayer cloned by nn.TransformerEncoder, nn.Linear), so state_dict keys, shapes and the seeded
default initialisation are identical; forward() runs on the HIP kernels (pnpp_hip/transformer.py).

Status: forward and backward are built, including the train-mode dropouts of
nn.TransformerEncoderLayer (own random streams: same distribution as the reference, not the same bits; parity with the
reference is checked in eval mode and in train mode with the dropout probabilities at 0, the dropout paths against a
float64 evaluation with the same masks).
"""
import torch
import torch.nn as nn


class PointTransformer(nn.Module):
    def __init__(self, in_dim=3, embed_dim=64, num_heads=4, depth=6):
        super().__init__()
        self.input_proj = nn.Linear(in_dim, embed_dim)
        encoder_layer = nn.TransformerEncoderLayer(d_model=embed_dim, nhead=num_heads, batch_first=True)
        self.transformer = nn.TransformerEncoder(encoder_layer, num_layers=depth)
        self.fc_out = nn.Linear(embed_dim, 3)
        self.num_heads = num_heads

    def forward(self, x: torch.Tensor):
        """x (B,N,3) -> (B,3)."""
        from pnpp_hip import transformer as T
        return T.point_transformer_forward(self, x)

    def set_dropout(self, p: float) -> "PointTransformer":
        """Sets the four dropout probabilities of every encoder layer (the reference's constructor leaves them at
        nn.TransformerEncoderLayer's default 0.1)."""
        for layer in self.transformer.layers:
            layer.dropout.p = layer.dropout1.p = layer.dropout2.p = float(p)
            layer.self_attn.dropout = float(p)
        return self
