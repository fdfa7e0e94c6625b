"""DATA INFRASTRUCTURE — extracts the merge table CLIP's tokenizer actually uses from the vocabulary file distributed with
CLIP (build container only; needs /root/reference):

    python oracle/make_bpe_table.py        ->  mvlpt_amd/data/bpe_merges.txt.gz   (48 894 lines "left right", ~0.4 MB)

`clip/simple_tokenizer.py:64-67` reads `bpe_simple_vocab_16e6.txt.gz`, drops the header line and keeps the first
49152 - 256 - 2 merges of its 262 k; only those define the 49 408-entry vocabulary.  The table is a constant of the published
tokenizer (data, like the token tables of oracle/make_token_tables.py); shipping it lets `mvlpt_amd.tokenizer.BPETokenizer`
tokenise arbitrary class lists on a box that has no copy of CLIP (the GPU box)."""
import gzip
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

N_MERGES = 49152 - 256 - 2
SRC = os.path.join(ref_shim.REFERENCE_ROOT, "clip", "bpe_simple_vocab_16e6.txt.gz")
OUT = os.path.join(ROOT, "mvlpt_amd", "data", "bpe_merges.txt.gz")

if __name__ == "__main__":
    with gzip.open(SRC, "rt", encoding="utf-8") as f:
        lines = f.read().split("\n")
    merges = lines[1:1 + N_MERGES]
    assert len(merges) == N_MERGES and all(len(m.split()) == 2 for m in merges)
    with gzip.GzipFile(OUT, "wb", compresslevel=9, mtime=0) as g:
        g.write(("\n".join(merges) + "\n").encode("utf-8"))
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB,", len(merges), "merges")
