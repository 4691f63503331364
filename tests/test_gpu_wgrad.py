"""GPU parity of the token-major weight-gradient GEMM (swg_linear_wgrad_bf16, the op behind every dW of the train
step) against fp32 torch on the same bf16-rounded operands.  Shapes cover the edges the kernel handles specially:
token counts below / not a multiple of the 32-token tile and of the slice length, outputs narrower than and not a
multiple of the 128 x 128 tile, padded leading dimensions with non-zero padding (must not leak into dW), += semantics.
Tolerance: operands are identical on both sides, so only the fp32 summation order differs: 2e-4 of max|dW|."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [  # rows, N, K, ldy, ldx
    (1, 4, 4, 8, 8),
    (16, 60, 60, 64, 64),
    (31, 96, 32, 96, 32),
    (33, 20, 132, 24, 136),
    (1000, 200, 96, 224, 96),
    (5120, 384, 96, 384, 96),
    (4097, 288, 96, 288, 128),
    (20000, 768, 260, 768, 288),
    (300001, 96, 96, 96, 96),
]


@pytest.mark.parametrize("rows,N,K,ldy,ldx", CASES)
def test_linear_wgrad_matches_torch(rows, N, K, ldy, ldx):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from swingnn_amd import _lib as L, engine as E
    g = torch.Generator(device="cpu").manual_seed(rows * 7 + N)
    dy = torch.randn(rows, ldy, generator=g).to(DEV).to(torch.bfloat16)
    x = torch.randn(rows, ldx, generator=g).to(DEV).to(torch.bfloat16)
    dw0 = torch.randn(N, K, generator=g).to(DEV)
    dw = dw0.clone()
    nbytes = L.lib.swg_linear_wgrad_workspace_bytes(N, K)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    for _ in range(2):  # two calls: the second accumulates on top of the first
        L.check(L.lib.swg_linear_wgrad_bf16(rows, N, K, dy.data_ptr(), ldy, x.data_ptr(), ldx, dw.data_ptr(),
                                            ws.data_ptr(), nbytes, E.stream_ptr()), "wgrad")
    torch.cuda.synchronize()
    ref = dy[:, :N].double().t() @ x[:, :K].double()
    got = (dw - dw0).double() / 2
    scale = float(ref.abs().max()) + 1e-6
    assert float((got - ref).abs().max()) <= 2e-4 * scale + 1e-5, (float((got - ref).abs().max()), scale)


def test_linear_wgrad_rejects_bad_arguments():
    from swingnn_amd import _lib as L, engine as E
    dy = torch.zeros(64, 96, device=DEV, dtype=torch.bfloat16)
    dw = torch.zeros(96, 96, device=DEV)
    ws = torch.empty(L.lib.swg_linear_wgrad_workspace_bytes(96, 96), dtype=torch.uint8, device=DEV)
    f = L.lib.swg_linear_wgrad_bf16
    assert f(64, 96, 96, None, 96, dy.data_ptr(), 96, dw.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr()) == -4
    assert f(64, 96, 96, dy.data_ptr(), 90, dy.data_ptr(), 96, dw.data_ptr(), ws.data_ptr(), ws.numel(), E.stream_ptr()) == -1
    assert f(4096, 96, 96, dy.data_ptr(), 96, dy.data_ptr(), 96, dw.data_ptr(), ws.data_ptr(), 1024, E.stream_ptr()) == -3
