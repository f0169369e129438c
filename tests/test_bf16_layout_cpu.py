"""CPU checks of the bf16 stream / buffer layout constants in csrc/bf16_common.h and bf16_weights.h: the segment starts
must be the running sums of (tiles x k-steps) in the order the kernels consume them, the streams must be whole 16-fragment
chunks (or padded up to one), and the training buffers' per-wave-block sizes must add up.  A silent edit of one constant
would otherwise only show as wrong numbers on the GPU."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf-tiny_amd", "csrc")


def _consts(path):
    txt = open(path).read()
    out = {}
    for name, val in re.findall(r"\b([A-Z][A-Z0-9_]*)\s*=\s*(\d+)\s*[,;]", txt):
        out[name] = int(val)
    return out, txt


def test_forward_stream_32x32x16():
    c, _ = _consts(os.path.join(CSRC, "bf16_common.h"))
    # point_info is folded into dir_info: the sigma head is a tile of its own on h7, dir_info reads gamma_d (2) + h7 (16)
    segs = [("BFS_L0", 8 * 4), ("BFS_L1", 3 * 8 * 16), ("BFS_L4", 8 * 20), ("BFS_L5", 3 * 8 * 16), ("BFS_SIG", 1 * 16),
            ("BFS_DIR", 4 * 18), ("BFS_COL", 1 * 8)]
    pos = 0
    for name, n in segs:
        assert c[name] == pos, name
        pos += n
    assert c["BF_NFRAG"] == pos == 1056 and pos % c["BF_CHUNK"] == 0
    # bias tiles: 8 layers x 8, sigma 1, dir 4, colour 1
    assert (c["BFB_L0"], c["BFB_SIGMA"], c["BFB_DIR"], c["BFB_COL"], c["BF_NBIAS_TILES"]) == (0, 64, 65, 69, 70)
    assert c["BF_NBIAS_TILES"] * 32 * 4 <= c["BF_BIAS_BYTES"]


def test_backward_stream():
    c, _ = _consts(os.path.join(CSRC, "bf16_common.h"))
    segs = [("BBS_COLT", 4 * 4), ("BBS_FOLDT", 8 * 9), ("BBS_L7T", 3 * 8 * 16), ("BBS_L4T", 8 * 16),
            ("BBS_L3T", 3 * 8 * 16), ("BBS_G0T", 2 * 32)]
    pos = 0
    for name, n in segs:
        assert c[name] == pos, name
        pos += n
    assert c["BBC_NFRAG"] == c["BBS_G0T"] == 984  # the coarse pass stops in front of the d gamma_p segments
    assert c["BBF_NFRAG"] == pos == 1048


def test_forward_stream_16x16x32():
    c, _ = _consts(os.path.join(CSRC, "bf16_weights.h"))  # (the 16x16x32 image's segment table lives beside its element function)
    segs = [("BXS_L0", 16 * 2), ("BXS_L1", 3 * 16 * 8), ("BXS_L4", 16 * 10), ("BXS_L5", 3 * 16 * 8), ("BXS_SIG", 1 * 8),
            ("BXS_DIR", 8 * 9), ("BXS_COL", 1 * 4)]
    pos = 0
    for name, n in segs:
        assert c[name] == pos, name
        pos += n
    assert c["BX_NFRAG"] == pos == 1044
    assert -(-pos // 16) * 16 <= 1056  # padded to whole chunks, it still fits the workspace region of the 32x32x16 image


def test_training_buffers():
    _, txt = _consts(os.path.join(CSRC, "bf16_common.h"))
    # bs_ks: gamma_p 4, h0..h7 16 each, c 8, gamma_d 2;  bg_ks: dpre0..7 16 each, dpre_dir 8, (dz, dspre) 2
    assert "t == BS_GP ? 4 : t < BS_C ? 16 : t == BS_C ? 8 : 2" in txt
    assert "t < BG_D ? 16 : t == BG_D ? 8 : 2" in txt
    assert 4 + 8 * 16 + 8 + 2 == 142 and 8 * 16 + 8 + 2 == 138
    assert "// 142 KiB per wave block" in txt and "// 138" in txt


def test_python_decoder_matches_header():
    """tests/test_gpu_bf16.py decodes the fragment layout with its own copies of the k-step tables"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("t_bf16", os.path.join(ROOT, "tests", "test_gpu_bf16.py"))
    src = open(spec.origin).read()
    assert "BS_KS = [4] + [16] * 8 + [8, 2]" in src and "BG_KS = [16] * 8 + [8, 2]" in src
