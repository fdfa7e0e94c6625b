"""Integer-exact check of the own BPE tokenizer against ids produced by the reference tokenizer: tests/golden/tokens.npz and
the four BASELINE class lists of mvlpt_amd/data/class_prompts.npz (4 442 prompts x 3 context sizes).  The merge table is
shipped (mvlpt_amd/data/bpe_merges.txt.gz, oracle/make_bpe_table.py), so these run everywhere — GPU box included."""
import os

import numpy as np
import pytest

from tests.golden_util import load_npz

VOCAB = None          # the shipped table (BPETokenizer's default)
CLIP_FILE = "/root/reference/clip/bpe_simple_vocab_16e6.txt.gz"


def test_shipped_merges_equal_clips_own_file():
    """Build container only: the shipped table is the 48 894 merges CLIP keeps of its vocabulary file, in order."""
    if not os.path.isfile(CLIP_FILE):
        pytest.skip("no copy of CLIP's vocabulary file here")
    from mvlpt_amd.tokenizer import BPETokenizer
    a, b = BPETokenizer(), BPETokenizer(CLIP_FILE)
    assert a.decoder == b.decoder and a.rank == b.rank and len(a.decoder) == 49408


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


def test_bpe_contract():
    from mvlpt_amd.tokenizer import BPETokenizer
    tok = BPETokenizer(VOCAB)
    t = tok.tokenize("a photo of a dog.", context_length=12)
    assert t.shape == (1, 12) and int(t[0, 0]) == 49406 and int(t.max()) == 49407
    with pytest.raises(RuntimeError):
        tok.tokenize("a very long sentence " * 30)
    assert tok.tokenize("a very long sentence " * 30, truncate=True)[0, -1] == 49407


def test_whole_class_lists_bit_exact_and_any_n_ctx():
    """f4 closed: every prompt of the four BASELINE class lists re-tokenised with the own tokenizer equals the reference
    tokenizer's table (ids, name lengths, CUT_CONTEXTLEN length) for n_ctx in {0, 4, 16}; other n_ctx go the same way."""
    import torch
    from mvlpt_amd import class_prompts as cp
    from mvlpt_amd.model import EOT_TOKEN, SOT_TOKEN, X_TOKEN
    z = cp._tables()
    for name in cp.LISTS:
        names = cp.class_names(name)
        for n_ctx in (0, 4, 16):
            ids, name_lens, cut = cp.tokenize_prompts(names, n_ctx, 77, tails=True)
            ref = z[f"{name}/ids_nctx{n_ctx}"].astype(np.int64)
            assert cut == ref.shape[1] and np.array_equal(ids[:, :cut].numpy(), ref) and not bool(ids[:, cut:].any()), (name, n_ctx)
            assert name_lens == z[f"{name}/name_lens"].astype(int).tolist()
        pre8, C = cp.load_class_prompts(name, 8)                      # a context size no table holds
        pre16, _ = cp.load_class_prompts(name, 16)
        i8, i16 = pre8.tokenized_prompts, pre16.tokenized_prompts
        assert i8.shape == (C, 77) and bool((i8[:, 0] == SOT_TOKEN).all()) and bool((i8[:, 1:9] == X_TOKEN).all())
        assert torch.equal(i8[:, 9:69], i16[:, 17:77]) and pre8.name_lens == pre16.name_lens      # same tail, 8 positions earlier
        assert torch.equal(i8.argmax(-1) + 8, i16.argmax(-1)) and int(i8.max()) == EOT_TOKEN
        cut8, _ = cp.load_class_prompts(name, 8, cut_contextlen=True)
        assert cut8.tokenized_prompts.shape[1] == cp.cut_context_length(name, 16) - 8
    with pytest.raises(ValueError):
        cp.load_class_prompts("caltech101", 75)
    # an arbitrary class list (not one of the shipped ones)
    ids, nl, cut = cp.tokenize_prompts(["golden retriever", "tabby_cat", "snoopy (cartoon beagle)"], 4)
    assert ids.shape == (3, 77) and nl[0] == 2 and cut == int(ids.argmax(-1).max()) + 1


def test_multitask_book_offsets():
    """trainers/mvlpt.py:585-645, 780-790: tasks in order, labels shifted by the class counts before them."""
    from mvlpt_amd import class_prompts as cp
    for name, n_tasks, total in (("coop11", 11, 2191), ("elevater20", 20, 1151)):
        b = cp.MultitaskBook.from_list(name)
        counts = cp.task_class_counts(name)
        assert len(b._task_names) == n_tasks and b.num_classes == total == len(b.classnames) == len(b.lab2cname)
        off = 0
        for i, (t, c) in enumerate(zip(b._task_names, counts)):
            assert b._id2task[i] == t and b._task2id[t] == i and b._task_class_idx[t] == (off, off + c) and len(b._labelmap[t]) == c
            assert b.global_label(i, 0) == off and b.global_label(t, c - 1) == off + c - 1
            assert b.lab2cname[off] == b._labelmap[t][0] == b.classnames[off]
            off += c
        with pytest.raises(IndexError):
            b.global_label(0, counts[0])
    b = cp.MultitaskBook.from_list("coop11")
    assert b._task_names[:2] == ["ImageNet", "Caltech101"] and b._task_class_idx["Caltech101"] == (1000, 1100)
    assert b.classnames[1000] == "accordion" and b.classnames[0] == "tench"
    b2 = cp.MultitaskBook([("a", ["x", "y"]), ("b", ["z"])])
    assert b2._task_class_idx == {"a": (0, 2), "b": (2, 3)} and b2.classnames == ["x", "y", "z"] and b2.lab2cname == {0: "x", 1: "y", 2: "z"}


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


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="re-reads the class NAME tables of /root/reference (build container)")
def test_class_prompt_tables_match_own_bpe():
    """Build container: the shipped tables (reference tokenizer) == our BPE on the same class lists, re-read from the
    reference's own name tables (elsewhere test_whole_class_lists_bit_exact_and_any_n_ctx does the same from the decoded names)."""
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
