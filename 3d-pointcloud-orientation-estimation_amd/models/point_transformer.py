"""models/point_transformer.py -- drop-in for the reference file of the same name (models/point_transformer.py:4-20).

The parameter containers are the reference's own modules in the same construction order (nn.Linear,
nn.TransformerEncoderLayer cloned by nn.TransformerEncoder, nn.Linear), so state_dict keys, shapes and the seeded
default initialisation are identical; forward() runs on the HIP kernels (pnpp_hip/transformer.py).

Status: the forward pass is built (eval-mode parity with the reference); the backward pass and
the train-mode dropouts of nn.TransformerEncoderLayer are not, so train-mode calls raise instead of falling back.
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
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("PointTransformer on the HIP path: only the forward pass is built so far "
                                      "(call .eval() / torch.no_grad()); there is no PyTorch fallback")
        return T.point_transformer_forward(self, x)
