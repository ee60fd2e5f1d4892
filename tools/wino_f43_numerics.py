#!/usr/bin/env python3
"""what would F(4x4,3x3) cost in accuracy on the K >= 256 BEV layers?  CPU experiment, no kernel: the Winograd
algorithm in fp32 (transforms and the 36 / 16 channel GEMMs all fp32, filters transformed in fp64 like the product) on bev_fusion-shaped
data against fp64 direct convolution, next to F(2x2,3x3) and to the direct fp32 convolution.  usage: wino_f43_numerics.py [Cin] [Cout] [S]"""
import sys
import torch
import torch.nn.functional as F

torch.manual_seed(0)
Cin, Cout, S = (int(a) for a in (sys.argv[1:4] + ["512", "512", "64"][len(sys.argv) - 1:]))

def mats(m):
    if m == 2:
        BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
        G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
        AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    else:                                                   # Lavin & Gray, F(4x4,3x3), points 0, +-1, +-2, inf
        BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                           [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
        G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                          [0, 0, 1]], dtype=torch.float64)
        AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
    return BT, G, AT

def wino(x, w, m):
    """x (1,Cin,S,S) fp32, w (Cout,Cin,3,3) fp32 -> y fp32; every product and sum in fp32 except G g G^T (fp64 -> fp32, as the product does)."""
    BT, G, AT = mats(m)
    a = m + 2
    U = (G @ w.double() @ G.T).float()                                        # (Cout,Cin,a,a)
    xp = F.pad(x, (1, 1, 1, 1))
    t = xp.unfold(2, a, m).unfold(3, a, m)                                    # (1,Cin,T,T,a,a)
    T = t.shape[2]
    V = BT.float() @ t @ BT.float().T                                         # fp32 input transform
    M = torch.einsum("ocij,ctuij->otuij", U, V[0])                            # a*a channel GEMMs, fp32 accumulate
    Y = AT.float() @ M @ AT.float().T                                         # (Cout,T,T,m,m)
    return Y.permute(0, 1, 3, 2, 4).reshape(1, w.shape[0], T * m, T * m)

x = torch.randn(1, Cin, S, S).clamp_(min=0)                                   # post-ReLU activations
w = torch.randn(Cout, Cin, 3, 3) * (1.0 / (9 * Cin)) ** 0.5
ref = F.conv2d(x.double(), w.double(), padding=1)
rel = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
print(f"Cin {Cin} Cout {Cout} map {S}x{S}: max |err| / max |y| against fp64 direct convolution")
print(f"  direct fp32 (torch CPU)      {rel(F.conv2d(x, w, padding=1)):.2e}")
print(f"  Winograd F(2x2,3x3) fp32     {rel(wino(x, w, 2)):.2e}   (16 multiplies per 4 outputs: 2.25x fewer than direct)")
print(f"  Winograd F(4x4,3x3) fp32     {rel(wino(x, w, 4)):.2e}   (36 multiplies per 16 outputs: 4x fewer than direct, 1.78x fewer than F(2x2))")
