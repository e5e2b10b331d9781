"""Byte-level BPE tokenizer of CLIP / open_clip (``open_clip.tokenize``, which ``FrozenOpenCLIPTtxtVisualEmbedder.forward`` calls
at tools/modules/clip_embedder.py:189) — host-side text preparation upstream of ``clip_text.ClipTextEngine``.

The algorithm is the published one (CLIP's ``simple_tokenizer``): lower-cased, whitespace-collapsed text is split by a fixed regular
expression, every piece is mapped byte-wise onto 256 printable code points, the last symbol gets the ``</w>`` suffix, and adjacent
symbol pairs are merged in the order of the merges file; ids index ``[256 byte symbols, 256 byte symbols + '</w>', merges...,
<start_of_text>, <end_of_text>]``.  ``tokenize`` frames each text as ``<start> ids <end>``, truncates to the context length keeping
``<end>`` last (open_clip's behaviour) and pads with zeros.

The MERGES FILE (``bpe_simple_vocab_16e6.txt.gz``, 49 152 - 256 - 2 merges) ships inside the open_clip package, which is not in this
image: pass its path (``bpe_path``).  Parity unpinned: there is no vocabulary here to compare ids against; ``tests/test_clip_cpu.py``
checks the algorithm on a synthetic merges file.  ``ftfy.fix_text`` (mojibake repair in the original's ``basic_clean``) is not
available and is skipped; ``html.unescape`` is applied twice as in the original.
"""
import gzip
import html
from functools import lru_cache
from typing import Dict, List, Sequence, Tuple, Union

import regex
import torch

SOT, EOT = "<start_of_text>", "<end_of_text>"


def _kept_bytes() -> List[int]:
    return list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))


@lru_cache()
def byte_symbols() -> Dict[int, str]:
    """byte -> printable code point: the printable Latin-1 ranges map to themselves, the other 68 bytes to 256, 257, ..."""
    table, extra = {b: chr(b) for b in _kept_bytes()}, 0
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _byte_symbol_order() -> List[str]:
    """Vocabulary order of the 256 byte symbols: the kept ranges first, then the remapped bytes in byte order."""
    sym, keep = byte_symbols(), _kept_bytes()
    rest = [b for b in range(256) if b not in set(keep)]
    return [sym[b] for b in keep + rest]


class ClipBpeTokenizer:
    def __init__(self, bpe_path: str, context_length: int = 77, vocab_size: int = 49408):
        opener = gzip.open if str(bpe_path).endswith(".gz") else open
        with opener(bpe_path, "rb") as f:
            lines = f.read().decode("utf-8").split("\n")
        n_merges = vocab_size - 512 - 2
        self.merges: List[Tuple[str, str]] = [tuple(ln.split()) for ln in lines[1:1 + n_merges] if len(ln.split()) == 2]      # (line 0 is a header)
        base = _byte_symbol_order()
        vocab = base + [s + "</w>" for s in base] + ["".join(m) for m in self.merges] + [SOT, EOT]
        self.ids = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(self.merges)}
        self.context_length = context_length
        self.sot, self.eot = self.ids[SOT], self.ids[EOT]
        self._memo: Dict[str, List[str]] = {}
        self._split = regex.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)

    def _merge(self, piece: str) -> List[str]:
        """piece: a regex match already mapped to byte symbols -> its BPE symbols."""
        if piece in self._memo:
            return self._memo[piece]
        word = list(piece[:-1]) + [piece[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for pair in zip(word[:-1], word[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            out, i = [], 0
            while i < len(word):                       # merge every occurrence of the best pair, left to right
                if i + 1 < len(word) and word[i] == best[0] and word[i + 1] == best[1]:
                    out.append(best[0] + best[1]); i += 2
                else:
                    out.append(word[i]); i += 1
            word = out
        self._memo[piece] = word
        return word

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(text)).strip()
        text = regex.sub(r"\s+", " ", text).strip().lower()
        sym = byte_symbols()
        ids: List[int] = []
        for m in self._split.findall(text):
            if m in (SOT, EOT):
                ids.append(self.ids[m])
                continue
            piece = "".join(sym[b] for b in m.encode("utf-8"))
            ids.extend(self.ids[s] for s in self._merge(piece))
        return ids

    def __call__(self, texts: Union[str, Sequence[str]]) -> torch.Tensor:
        """-> int64 [B, context_length]: <start> ids <end>, zero padded; long texts are cut and end with <end>."""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), self.context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > self.context_length:
                ids = ids[: self.context_length]
                ids[-1] = self.eot
            out[i, : len(ids)] = torch.tensor(ids, dtype=torch.long)
        return out
