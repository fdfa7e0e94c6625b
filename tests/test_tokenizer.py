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


# ---------------------------------------------------------------------------------------------- shipped token tables
def test_class_prompt_tables_structure():
    import torch
    """mvlpt_amd/data/class_prompts.npz (reference tokenizer output, integers): structure every table must have — runs
    everywhere, no merge table needed."""
    from mvlpt_amd.class_prompts import LISTS, cut_context_length, load_class_prompts, task_class_counts
    from mvlpt_amd.model import EOT_TOKEN, SOT_TOKEN, X_TOKEN
    want_classes = {"caltech101": 100, "imagenet1k": 1000, "coop11": 2191, "elevater20": 1151}
    want_cut = {("caltech101", 16): 24, ("imagenet1k", 16): 30, ("elevater20", 4): 23}
    for name in LISTS:
        assert sum(task_class_counts(name)) == want_classes[name]
        for n_ctx in (0, 4, 16):
            pre, C = load_class_prompts(name, n_ctx)
            ids = pre.tokenized_prompts
            assert C == want_classes[name] and ids.shape == (C, 77) and ids.dtype == torch_long()
            assert bool((ids[:, 0] == SOT_TOKEN).all()) and bool((ids.max(dim=1).values == EOT_TOKEN).all())
            eot = ids.argmax(dim=-1)
            if n_ctx:
                assert bool((ids[:, 1:1 + n_ctx] == X_TOKEN).all())
                # EOT index = n_ctx + name_len + 2 (SOT, ctx, name, '.', EOT) — except where the BPE
                # merges the name's trailing punctuation with the final '.' ("snoopy (cartoon beagle)." -> ").": one
                # token less); the reference takes the EOT position from argmax(tokenized_prompts), as we do
                want = torch.tensor([n_ctx + nl + 2 for nl in pre.name_lens])
                assert bool(((eot == want) | (eot == want - 1)).all()) and float((eot == want).float().mean()) > 0.97
            assert bool((ids.gather(1, (eot + 1).clamp(max=76).unsqueeze(1))[eot < 76] == 0).all())    # zero padding after EOT
            cut, _ = load_class_prompts(name, n_ctx, cut_contextlen=True)
            assert cut.tokenized_prompts.shape[1] == int(eot.max()) + 1 == cut_context_length(name, n_ctx)
            if (name, n_ctx) in want_cut:
                assert cut.tokenized_prompts.shape[1] == want_cut[(name, n_ctx)]


def torch_long():
    import torch
    return torch.long


@pytest.mark.skipif(VOCAB is None, reason="CLIP BPE merge table not available")
def test_class_prompt_tables_match_own_bpe():
    """Where the merge table exists: the shipped tables (reference tokenizer) == our BPE on the same class lists, which
    are re-read from the reference's own name tables."""
    import torch
    from mvlpt_amd.class_prompts import load_class_prompts
    from mvlpt_amd.tokenizer import BPETokenizer
    from oracle.make_token_tables import class_lists
    tok = BPETokenizer(VOCAB)
    for name, tasks in class_lists().items():
        names = [n.replace("_", " ") for _, cl in tasks for n in cl]
        for n_ctx in (0, 16):
            prefix = " ".join(["X"] * n_ctx) if n_ctx else "a photo of a "
            pre, _ = load_class_prompts(name, n_ctx)
            step = max(1, len(names) // 150)                 # every class of the small lists, a stride of the big ones
            ours = tok.tokenize([prefix + " " + n + "." for n in names[::step]])
            assert torch.equal(ours, pre.tokenized_prompts[::step]), (name, n_ctx)
            assert [len(tok.encode(n)) for n in names[::step]] == pre.name_lens[::step]
