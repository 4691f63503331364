"""Pin oracle/codec_ref.py (attribute codecs) against fixtures produced by the reference's own
utils/attribute_code.attribute_converter (tests/golden/make_golden_codec.py).  Bit-exact: integer / +-1 outputs."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import codec_ref as CR
from oracle import filler
from tests.golden.cases import CODEC_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden_codec():
    with np.load(os.path.join(ROOT, "tests", "golden", "golden_codec.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def codec_inputs(name, is_adj, enc=None):
    """(flags, ints, raw sample in `enc`) exactly as make_golden_codec.py builds them."""
    c = CODEC_CASES[name]
    B, N, k, seed = c["B"], c["N"], c["k"], c["seed"]
    flags = filler.node_flags("flags", B, N, c["n_lo"], c["n_hi"], seed)
    tag = f"{name}/{'adj' if is_adj else 'node'}"
    shape = (B, N, N) if is_adj else (B, N)
    ints = ((filler.uniform(tag + "/int", shape, seed) + 1.0) * 0.5 * k).floor().clamp(0, k - 1)
    ints = (CR.mask_adjs if is_adj else CR.mask_nodes)(ints, flags)
    raw = None
    if enc is not None:
        C = {"ddpm": 1, "one_hot": k, "bits": int(math.ceil(math.log2(k)))}[enc]
        if enc == "ddpm":
            raw = filler.normalish(tag + "/raw_ddpm", shape, seed) * 0.9
        elif is_adj:
            raw = filler.normalish(tag + f"/raw_{enc}", (B, C, N, N), seed) * 0.9
            raw = raw + raw.transpose(-1, -2)
        else:
            raw = filler.normalish(tag + f"/raw_{enc}", (B, N, C), seed) * 0.9
    return flags, ints, raw, k, tag


@pytest.mark.parametrize("name", list(CODEC_CASES))
@pytest.mark.parametrize("is_adj", [False, True])
def test_codec_oracle_matches_reference(golden_codec, name, is_adj):
    for enc, to_enc, to_int in (("ddpm", CR.int_to_ddpm, lambda e, f, k: CR.ddpm_to_int(e, f, k)),
                                ("one_hot", CR.int_to_one_hot, CR.one_hot_to_int),
                                ("bits", CR.int_to_bits, CR.bits_to_int)):
        flags, ints, raw, k, tag = codec_inputs(name, is_adj, enc)
        e = to_enc(ints, flags, k)
        assert np.array_equal(e.numpy().astype(np.float32), golden_codec[f"{tag}/enc_{enc}"]), (tag, enc)
        assert torch.equal(to_int(e, flags, k).float(), ints), (tag, enc)            # round trip
        d = CR.decode_samples(raw, flags, enc, k, is_adj)
        assert np.array_equal(d.numpy().astype(np.float32), golden_codec[f"{tag}/dec_{enc}"]), (tag, enc)
