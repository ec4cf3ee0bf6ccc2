"""Caption <-> token-id conversion either side of the path -- SURVEY.md section 8(f) rows 1 and 4.

The reference tokenises captions with `DistilBertTokenizer(text=..., padding='max_length', truncation=True, max_length=MAX_LENGTH)`
(ref CLIP-DDPM.py:181-182, 205) and turns sampled ids back into strings with `tokenizer.decode(index)` (ref :594, :602, :623) before
BLEU; its 16-dim ablation uses a character-level `DictTokenizer` (ref :153-165, :184-188).  The vocabulary file itself is a download
(ref :40-50) that is not in the reference repo, so this module takes the vocabulary (a `vocab.txt` path, a list or a dict) from the caller
and restates the two algorithms:

  * `WordPiece`      -- BERT uncased: normaliser (clean text, CJK spacing, lower-case, strip accents), whitespace + punctuation
                        pre-tokeniser, greedy longest-match WordPiece with the "##" continuation prefix, [CLS] .. [SEP] template,
                        truncation and [PAD] padding; `decode` = the WordPiece decoder with its per-token clean-up, special tokens kept
                        (the reference passes no `skip_special_tokens`, which is why its BLEU references are wrapped in "[CLS] .. [SEP]").
  * `DictTokenizer`  -- ref :153-165 / :184-188: one id per CHARACTER of the caption, 0 / 1 as begin / end markers, 'UNK' padding.

Pinned in tests/test_next_rows.py against the `tokenizers` / `transformers` implementation the reference calls (installed in this image),
built on a synthetic vocabulary.  Host-side only: nothing here touches the GPU.
"""
from __future__ import annotations

import unicodedata
from typing import Iterable, Sequence

_CLEANUP = ((" .", "."), (" ?", "?"), (" !", "!"), (" ,", ","), (" ' ", "'"), (" n't", "n't"), (" 'm", "'m"), (" do not", " don't"),
            (" 's", "'s"), (" 've", "'ve"), (" 're", "'re"))


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F
            or 0x2B920 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPiece:
    """BERT-uncased WordPiece over a caller-supplied vocabulary.  `tok(text, max_length=16)` -> {"input_ids", "attention_mask"} lists."""

    def __init__(self, vocab, unk="[UNK]", sep="[SEP]", pad="[PAD]", cls="[CLS]", mask="[MASK]", lowercase=True, max_chars_per_word=100):
        if isinstance(vocab, str):
            with open(vocab, encoding="utf-8") as f:
                vocab = [ln.rstrip("\n") for ln in f]
        if not isinstance(vocab, dict):
            vocab = {t: i for i, t in enumerate(vocab)}
        self.vocab = dict(vocab)
        self.inv = {i: t for t, i in self.vocab.items()}
        self.unk, self.sep, self.pad, self.cls, self.mask = unk, sep, pad, cls, mask
        self.special = {unk, sep, pad, cls, mask}
        for s in (unk, sep, pad, cls):
            assert s in self.vocab, f"vocabulary lacks {s}"
        self.lowercase = lowercase
        self.max_chars = max_chars_per_word

    @property
    def vocab_size(self):
        return len(self.vocab)

    # ---- text -> ids
    def normalize(self, text):
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_whitespace(ch):
                out.append(" ")
            elif _is_cjk(cp):
                out.extend((" ", ch, " "))
            else:
                out.append(ch)
        text = "".join(out)
        if self.lowercase:
            text = "".join(c for c in unicodedata.normalize("NFD", text) if unicodedata.category(c) != "Mn").lower()
        return text

    def words(self, text):
        """Whitespace split, every punctuation character its own word."""
        res, cur = [], []
        for ch in self.normalize(text):
            if _is_whitespace(ch):
                if cur:
                    res.append("".join(cur))
                    cur = []
            elif _is_punctuation(ch):
                if cur:
                    res.append("".join(cur))
                    cur = []
                res.append(ch)
            else:
                cur.append(ch)
        if cur:
            res.append("".join(cur))
        return res

    def pieces(self, word):
        if len(word) > self.max_chars:
            return [self.unk]
        out, a = [], 0
        while a < len(word):
            b, hit = len(word), None
            while a < b:
                sub = ("##" if a else "") + word[a:b]
                if sub in self.vocab:
                    hit = sub
                    break
                b -= 1
            if hit is None:
                return [self.unk]
            out.append(hit)
            a = b
        return out

    def tokenize(self, text):
        # a special token written out in the text (the reference's BLEU targets are "[CLS] caption [SEP]") stays one token
        toks, rest = [], text
        while rest:
            at, which = min(((rest.find(s), s) for s in self.special if s in rest), default=(-1, None))
            if which is None:
                break
            for w in self.words(rest[:at]):
                toks.extend(self.pieces(w))
            toks.append(which)
            rest = rest[at + len(which):]
        for w in self.words(rest):
            toks.extend(self.pieces(w))
        return toks

    def __call__(self, text, max_length=None, padding="max_length", truncation=True):
        ids = [self.vocab[t] for t in self.tokenize(text)]
        if truncation and max_length is not None:
            ids = ids[:max(max_length - 2, 0)]
        ids = [self.vocab[self.cls]] + ids + [self.vocab[self.sep]]
        mask = [1] * len(ids)
        if padding == "max_length" and max_length is not None and len(ids) < max_length:
            n = max_length - len(ids)
            ids, mask = ids + [self.vocab[self.pad]] * n, mask + [0] * n
        return {"input_ids": ids, "attention_mask": mask}

    def encode_batch(self, texts: Iterable[str], max_length):
        """-> (input_ids [n][max_length], attention_mask [n][max_length]) for data.ClipCaptionDataset."""
        enc = [self(t, max_length=max_length) for t in texts]
        return [e["input_ids"] for e in enc], [e["attention_mask"] for e in enc]

    # ---- ids -> text
    def decode(self, ids: Sequence, skip_special_tokens=False, cleanup="token"):
        """cleanup="token" (default): the clean-up table is applied to every token on its own, which is what the tokenizers-backed
        `DistilBertTokenizer` of transformers 5.x does (the installed 5.15: identical strings on a synthetic vocabulary, tests/test_next_rows.py) --
        with a leading space in front of each token " ' " and " do not" can never match.  cleanup="string": the table runs over the JOINED text, the
        behaviour of the slow `DistilBertTokenizer` of the reference's era (transformers 4.2x, `clean_up_tokenization` on the whole string):
        "man ' s" -> "man's", "do not" -> "don't".  The reference's BLEU targets are raw `[CLS] caption [SEP]` strings (ref :625-627) split on
        spaces, so candidates with apostrophes tokenise differently under the two: use "string" to reproduce the reference stack's BLEU, "token"
        to match the installed transformers.  cleanup=None: no clean-up."""
        if cleanup not in ("token", "string", None):
            raise ValueError(f"cleanup must be 'token', 'string' or None, not {cleanup!r}")
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        toks = [self.inv.get(int(i), self.unk) for i in ids]
        if skip_special_tokens:
            toks = [t for t in toks if t not in self.special]
        out = []
        for k, t in enumerate(toks):
            if k:
                t = t[2:] if t.startswith("##") else " " + t
            if cleanup == "token":
                for a, b in _CLEANUP:
                    t = t.replace(a, b)
            out.append(t)
        text = "".join(out)
        if cleanup == "string":
            for a, b in _CLEANUP:
                text = text.replace(a, b)
        return text


class DictTokenizer:
    """The reference's character-level ablation tokenizer (ref :153-165, :184-188): ids of the caption's first MAX_LENGTH-2 characters
    between 0 and 1, padded with the 'UNK' id; decode joins the dictionary keys with spaces."""

    def __init__(self, dictionary: dict):
        self.dictionary = dict(dictionary)
        self.inv = {}
        for k, v in self.dictionary.items():
            self.inv.setdefault(v, k)                              # `.index(i)` of the reference finds the FIRST key with that value

    def __len__(self):
        return len(self.dictionary)

    def __getitem__(self, k):
        return self.dictionary[k]

    def __call__(self, caption, max_length):
        unk = self.dictionary["UNK"]
        ids = [0] + [self.dictionary.get(c, unk) for c in caption[:max_length - 2]] + [1]
        pad = max(0, max_length - len(ids))
        return {"input_ids": ids + [unk] * pad, "attention_mask": [1] * len(ids) + [0] * pad}

    def decode(self, index):
        index = index.tolist() if hasattr(index, "tolist") else list(index)
        return " ".join(self.inv[int(i)] for i in index)
