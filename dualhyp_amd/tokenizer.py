"""Tokenizers for the harnesses (inference/ger.py:194-201, finetune/ger.py:88-90).

The reference uses `AutoTokenizer.from_pretrained(checkpoint_dir, use_fast=True)` (HF `LlamaTokenizerFast`)
and only three things of it: `encode(str) -> ids` (BOS prepended), `decode(ids) -> str`, `eos_token` /
`eos_token_id`.  `load_tokenizer` returns that HF object when the checkpoint directory holds tokenizer files;
`ByteTokenizer` is a dependency-free stand-in with the same three members for runs without a checkpoint
(smoke tests, synthetic weights): BOS = 1, EOS = 2, one id per UTF-8 byte (byte + 3), vocabulary 259.
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable, List, Union


RELIABILITY_TOKENS = ("<<C>>", "<<M>>", "<<N>>")     # inference/relprompt.py:341


class ByteTokenizer:
    bos_token_id, eos_token_id, pad_token_id = 1, 2, 0
    eos_token = "</s>"
    vocab_size = 259

    def __init__(self) -> None:
        self.special = {self.eos_token: self.eos_token_id}

    def add_reliability_tokens(self, first_id: int) -> None:
        """<<C>>/<<M>>/<<N>> -> first_id.. (the rows `resize_token_embeddings(3)` appends to wte)."""
        for i, tok in enumerate(RELIABILITY_TOKENS):
            self.special[tok] = first_id + i

    def encode(self, text: str) -> List[int]:
        ids, rest = [self.bos_token_id], text
        while rest:                                   # special strings in the text become their ids, as with HF
            hits = [(rest.find(s), s) for s in self.special if rest.find(s) >= 0]
            if not hits:
                ids += [b + 3 for b in rest.encode("utf-8")]
                break
            cut, tok = min(hits)
            ids += [b + 3 for b in rest[:cut].encode("utf-8")] + [self.special[tok]]
            rest = rest[cut + len(tok):]
        return ids

    def decode(self, ids: Iterable[int]) -> str:
        """Special tokens are skipped, like `decode(..., skip_special_tokens=True)`; the reference's
        `output[len(tokenizer.decode(encoded)):]` slicing then works on plain text."""
        return bytes(int(i) - 3 for i in ids if 3 <= int(i) < 259).decode("utf-8", errors="replace")


class _HFTokenizer:
    """The three members the harness needs over a Hugging Face tokenizer."""

    def __init__(self, tok) -> None:
        self.tok = tok
        if tok.pad_token is None:                     # inference/ger.py:200-201
            tok.pad_token = tok.eos_token
        self.eos_token, self.eos_token_id = tok.eos_token, tok.eos_token_id

    def add_reliability_tokens(self, first_id: int) -> None:
        self.tok.add_special_tokens({"additional_special_tokens": list(RELIABILITY_TOKENS)})
        got = self.tok.convert_tokens_to_ids(list(RELIABILITY_TOKENS))
        if got != [first_id, first_id + 1, first_id + 2]:
            raise ValueError(f"reliability tokens got ids {got}, the decoder's added wte rows are {first_id}..{first_id + 2}")

    def encode(self, text: str) -> List[int]:
        return list(self.tok.encode(text))

    def decode(self, ids) -> str:
        return self.tok.decode([int(i) for i in ids])


def load_tokenizer(checkpoint_dir: Union[str, Path], kind: str = "auto"):
    """kind: 'hf' (tokenizer files in `checkpoint_dir`), 'byte', or 'auto' (hf when the files are there)."""
    d = Path(checkpoint_dir)
    has_files = any((d / n).is_file() for n in ("tokenizer.json", "tokenizer.model", "tokenizer_config.json"))
    if kind == "byte" or (kind == "auto" and not has_files):
        if kind == "auto":
            print(f"[dualhyp_amd] no tokenizer files in {str(d)!r}: using the byte-level stand-in tokenizer")
        return ByteTokenizer()
    if not has_files:
        raise FileNotFoundError(f"{str(d)!r} holds no tokenizer.json / tokenizer.model")
    from transformers import AutoTokenizer
    return _HFTokenizer(AutoTokenizer.from_pretrained(str(d), use_fast=True))
