"""CPU-side checks of the C ABI: libuig.so loads, exports every symbol include/uig.h declares, the ctypes table covers
them all, and argument validation returns errors without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "uig.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uig_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    import unpaired_image_generation_amd as u
    lib = u.lib.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in uig.h but not exported by libuig.so"
        assert n in u.lib.SIGNATURES, f"{n} has no ctypes signature in lib.py"
    assert set(u.lib.SIGNATURES) == set(names)
    assert b"gfx950" in lib.uig_version()


def test_argument_errors_do_not_launch():
    import unpaired_image_generation_amd as u
    lib = u.lib.lib()
    # null pointers / bad shapes are rejected before any HIP call
    rc = lib.uig_conv_gather(None, None, None, None, 1, 8, 8, 8, 8, 3, 3, 1, 1, 0, 0, 8, 8, 8, 8, 0, 0.0, 0, None)
    assert rc < 0 and b"null" in lib.uig_last_error()
    buf = (ctypes.c_float * 16)()
    p = ctypes.addressof(buf)
    rc = lib.uig_conv_gather(p, p, None, p, 1, 8, 8, 3, 8, 3, 3, 1, 1, 0, 0, 8, 8, 8, 8, 0, 0.0, 0, None)
    assert rc < 0 and b"multiple of 8" in lib.uig_last_error()
    rc = lib.uig_conv_gather(p, p, None, p, 1, 8, 8, 8, 8, 3, 3, 1, 1, 0, 0, 9, 9, 8, 8, 0, 0.0, 0, None)
    assert rc < 0 and b"does not match" in lib.uig_last_error()
    rc = lib.uig_instnorm_act_fwd(p, None, p, p, p, 1, 16, 12, 1e-5, 0, 0.0, 0, None)
    assert rc < 0
    with pytest.raises(RuntimeError):
        u.lib.check(rc, "uig_instnorm_act_fwd")
    assert lib.uig_wgrad_workspace_bytes(256, 256, 3, 3, 4) == 4 * 256 * 9 * 256 * 4


def test_state_dict_layout_matches_oracle():
    """Module surface: same keys / shapes as the stock-torch restatement (Appendix A), built without a GPU."""
    import torch
    import unpaired_image_generation_amd as u
    from oracle.torch_oracle import Discriminator as OD, Generator as OG
    g = u.Generator(n_blocks=6, device="cpu"); d = u.Discriminator(device="cpu")
    og, od = OG(n_blocks=6), OD()
    assert {k: tuple(v.shape) for k, v in g.state_dict().items()} == {k: tuple(v.shape) for k, v in og.state_dict().items()}
    assert {k: tuple(v.shape) for k, v in d.state_dict().items()} == {k: tuple(v.shape) for k, v in od.state_dict().items()}
    g.load_state_dict(og.state_dict())
    assert torch.equal(g.state_dict()["10.b.5.weight"], og.state_dict()["10.b.5.weight"])


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "unpaired-image-generation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("oracle module layout", ""), f"{f} mentions the oracle"
