"""Byte-level BPE tokenizer for CLIP prompts (row f4 of SURVEY.md §8: init-time, CPU, integer-exact).

Own implementation of the published algorithm the reference uses (clip/simple_tokenizer.py:62-132 and
`clip.tokenize`, clip/clip.py:187-223): text -> lower-case, whitespace-collapsed -> regex word split -> UTF-8
bytes mapped to printable code points -> greedy lowest-rank pair merging -> vocabulary ids; SOT = 49406, EOT = 49407.
The merge table is DATA shipped with the package: `mvlpt_amd/data/bpe_merges.txt.gz` = the 48 894 merges CLIP keeps of its
`bpe_simple_vocab_16e6.txt.gz` (oracle/make_bpe_table.py); CLIP's own file is accepted too (path argument or MVLPT_BPE_VOCAB).
Parity pin: tests/golden/tokens.npz and the four class-list tables of mvlpt_amd/data/class_prompts.npz, all generated with
the reference tokenizer.
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache
from typing import Dict, List, Sequence, Tuple, Union

import regex
import torch

N_MERGES = 49152 - 256 - 2        # merges kept by CLIP (vocabulary 49408 = 256 + 256 + merges + 2 specials)
SOT, EOT = "<|startoftext|>", "<|endoftext|>"
_WORD = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                      regex.IGNORECASE)


@lru_cache()
def _byte_symbols() -> Dict[int, str]:
    """Reversible byte -> printable unicode symbol table (GPT-2 byte-level BPE).  The ORDER of this dict is the
    order of the first 256 vocabulary ids: the 188 printable latin-1 bytes ('!'..'~', '¡'..'¬', '®'..'ÿ') map to
    themselves and come first, the remaining 68 bytes map to code points 256, 257, … and follow."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table = {b: chr(b) for b in keep}
    extra = 0
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table


SHIPPED_MERGES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bpe_merges.txt.gz")


class BPETokenizer:
    def __init__(self, vocab_path: str | None = None):
        vocab_path = vocab_path or os.environ.get("MVLPT_BPE_VOCAB") or SHIPPED_MERGES
        if not os.path.isfile(vocab_path):
            raise FileNotFoundError(f"BPE merge table not found: {vocab_path} (shipped: {SHIPPED_MERGES}; CLIP's own "
                                    "bpe_simple_vocab_16e6.txt.gz works too: pass its path or set MVLPT_BPE_VOCAB)")
        with gzip.open(vocab_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        if lines and "#version" in lines[0]:
            lines = lines[1:]                 # CLIP's file opens with a version header; the shipped table has none
        merges: List[Tuple[str, str]] = [tuple(l.split()) for l in lines[:N_MERGES]]
        if len(merges) != N_MERGES or any(len(m) != 2 for m in merges):
            raise ValueError(f"{vocab_path}: expected {N_MERGES} merges of two symbols")
        symbols = list(_byte_symbols().values())
        vocab = symbols + [s + "</w>" for s in symbols] + ["".join(m) for m in merges] + [SOT, EOT]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = vocab
        self.rank = {m: i for i, m in enumerate(merges)}
        self._cache: Dict[str, List[str]] = {}

    # ---- one word -> BPE units
    def _merge_word(self, word: str) -> List[str]:
        hit = self._cache.get(word)
        if hit is not None:
            return hit
        units = list(word[:-1]) + [word[-1] + "</w>"]
        while len(units) > 1:
            best, best_rank = -1, None
            for i in range(len(units) - 1):                     # lowest-rank adjacent pair
                r = self.rank.get((units[i], units[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best < 0:
                break
            a, b = units[best], units[best + 1]
            out, i = [], 0
            while i < len(units):                               # merge every occurrence of (a, b), left to right
                if i + 1 < len(units) and units[i] == a and units[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(units[i])
                    i += 1
            units = out
        self._cache[word] = units
        return units

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(text)).strip()       # ftfy.fix_text is the identity on clean ASCII names
        text = regex.sub(r"\s+", " ", text).strip().lower()
        sym = _byte_symbols()
        ids: List[int] = []
        for w in _WORD.findall(text):
            mapped = "".join(sym[b] for b in w.encode("utf-8"))
            ids.extend(self.encoder[u] for u in self._merge_word(mapped))
        return ids

    def decode(self, ids: Sequence[int]) -> str:
        """Inverse of `encode` up to whitespace / case (clip/simple_tokenizer.py:129-132): word ends become single spaces."""
        rev = {c: b for b, c in _byte_symbols().items()}
        out = bytearray()
        for i in ids:
            piece = self.decoder[int(i)]
            if piece in (SOT, EOT):
                out += piece.encode() + b" "
                continue
            end = piece.endswith("</w>")
            out += bytes(rev[c] for c in (piece[:-4] if end else piece))
            if end:
                out += b" "
        return out.decode("utf-8", errors="replace")

    def tokenize(self, texts: Union[str, Sequence[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        """`clip.tokenize` contract: LongTensor [n, context_length], [SOT] + ids + [EOT], zero padded."""
        if isinstance(texts, str):
            texts = [texts]
        sot, eot = self.encoder[SOT], self.encoder[EOT]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [sot] + self.encode(t) + [eot]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = eot
            out[i, :len(ids)] = torch.tensor(ids)
        return out
