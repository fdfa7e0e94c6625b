"""Integer-exact check of the own BPE tokenizer against ids produced by the reference tokenizer
(tests/golden/tokens.npz).  Needs CLIP's merge table, which is data we do not ship: runs where it is available
(build container: under /root/reference, or $MVLPT_BPE_VOCAB) and is skipped elsewhere."""
import os

import numpy as np
import pytest

from tests.golden_util import load_npz

CANDIDATES = [os.environ.get("MVLPT_BPE_VOCAB", ""), "/root/reference/clip/bpe_simple_vocab_16e6.txt.gz"]
VOCAB = next((p for p in CANDIDATES if p and os.path.isfile(p)), None)


@pytest.mark.skipif(VOCAB is None, reason="CLIP BPE merge table not available")
def test_bpe_ids_bit_exact_vs_reference():
    from mvlpt_amd.tokenizer import BPETokenizer
    tok = BPETokenizer(VOCAB)
    z = load_npz("tokens")
    names = [str(n) for n in z["names"]]
    assert [len(tok.encode(n)) for n in names] == z["name_lens"].tolist()
    for n_ctx in (0, 4, 16):
        prefix = " ".join(["X"] * n_ctx) if n_ctx else "a photo of a "
        ids = tok.tokenize([prefix + " " + n + "." for n in names]).numpy()
        assert np.array_equal(ids, z[f"ids_nctx{n_ctx}"]), f"n_ctx={n_ctx}"
        assert max(len(tok.encode(prefix + " " + n + ".")) + 2 for n in names) == int(z[f"cutlen_nctx{n_ctx}"])


@pytest.mark.skipif(VOCAB is None, reason="CLIP BPE merge table not available")
def test_bpe_contract():
    from mvlpt_amd.tokenizer import BPETokenizer
    tok = BPETokenizer(VOCAB)
    t = tok.tokenize("a photo of a dog.", context_length=12)
    assert t.shape == (1, 12) and int(t[0, 0]) == 49406 and int(t.max()) == 49407
    with pytest.raises(RuntimeError):
        tok.tokenize("a very long sentence " * 30)
    assert tok.tokenize("a very long sentence " * 30, truncate=True)[0, -1] == 49407


def test_missing_vocab_is_loud():
    from mvlpt_amd.tokenizer import BPETokenizer
    with pytest.raises(FileNotFoundError):
        BPETokenizer("/nonexistent/bpe.txt.gz")
