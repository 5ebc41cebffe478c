"""Host-side text splitter with the interface of the reference's tokenization module.

Behaviour follows /root/reference/tokenization.py:161-399 (the BERT "basic + wordpiece"
scheme the loader calls through ``FullTokenizer.tokenize``, data_loader_json.py:54,899):

  basic pass      drop NUL / U+FFFD / control characters, map every whitespace character to a
                  blank (:286-297); put blanks around CJK ideographs (:251-284); split on
                  blanks; with do_lower_case lower() then NFD and drop category Mn (:212-229);
                  every punctuation character becomes a token of its own (:231-249, :386-399)
  wordpiece pass  greedy longest match against the vocabulary, continuation pieces carry the
                  '##' prefix; a word with no full cover, or longer than 200 characters,
                  becomes '[UNK]' (:300-359)

Pinned by tests/golden/tokenizer_cases.json (outputs of the reference's own tokenizer).
The implementation is table driven: character classes are cached per code point and the
punctuation split is one regular-expression-free scan over precomputed flags.
"""
import collections
import unicodedata

_CJK_RANGES = ((0x4E00, 0x9FFF), (0x3400, 0x4DBF), (0x20000, 0x2A6DF), (0x2A700, 0x2B73F),
               (0x2B740, 0x2B81F), (0x2B820, 0x2CEAF), (0xF900, 0xFAFF), (0x2F800, 0x2FA1F))

# character classes of the basic pass
_KEEP, _DROP, _BLANK, _PUNCT, _CJK = range(5)
_class_cache = {}


def _char_class(ch):
    k = _class_cache.get(ch)
    if k is not None:
        return k
    cp = ord(ch)
    cat = unicodedata.category(ch)
    if ch in " \t\n\r" or cat == "Zs":
        k = _BLANK
    elif cp == 0 or cp == 0xFFFD or cat[0] == "C":
        k = _DROP
    elif any(lo <= cp <= hi for lo, hi in _CJK_RANGES):
        k = _CJK
    elif 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126 or cat[0] == "P":
        k = _PUNCT
    else:
        k = _KEEP
    _class_cache[ch] = k
    return k


def convert_to_unicode(text):
    if isinstance(text, str):
        return text
    if isinstance(text, bytes):
        return text.decode("utf-8", "ignore")
    raise ValueError("Unsupported string type: %s" % (type(text)))


def load_vocab(vocab_file):
    """token -> line index, in file order; a repeated token keeps its LAST index (as a dict
    assignment loop does, tokenization.py:121-133)."""
    vocab = collections.OrderedDict()
    with open(vocab_file, "r", encoding="utf-8") as reader:
        for index, line in enumerate(reader):
            vocab[line.strip()] = index
    return vocab


def whitespace_tokenize(text):
    return text.split()


def _strip_accents(word):
    return "".join(c for c in unicodedata.normalize("NFD", word) if unicodedata.category(c) != "Mn")


def _split_punct(word, out):
    start = -1
    for i, ch in enumerate(word):
        if _char_class(ch) == _PUNCT:
            if start >= 0:
                out.append(word[start:i])
                start = -1
            out.append(ch)
        elif start < 0:
            start = i
    if start >= 0:
        out.append(word[start:])


class BasicTokenizer(object):
    def __init__(self, do_lower_case=True):
        self.do_lower_case = do_lower_case

    def tokenize(self, text):
        text = convert_to_unicode(text)
        buf = []
        for ch in text:
            k = _char_class(ch)
            if k == _DROP:
                continue
            if k == _BLANK:
                buf.append(" ")
            elif k == _CJK:
                buf.append(" " + ch + " ")
            else:
                buf.append(ch)
        pieces = []
        for word in "".join(buf).split():
            if self.do_lower_case:
                word = _strip_accents(word.lower())
            _split_punct(word, pieces)
        # lower()/NFD can surface blanks inside a piece: split once more, as the reference's
        # final join + whitespace split does (:217)
        return " ".join(pieces).split()


class WordpieceTokenizer(object):
    def __init__(self, vocab, unk_token="[UNK]", max_input_chars_per_word=200):
        self.vocab = vocab
        self.unk_token = unk_token
        self.max_input_chars_per_word = max_input_chars_per_word

    def _cover(self, word):
        """pieces of the greedy longest-match cover, or None when some position has no match"""
        pieces, start, n = [], 0, len(word)
        while start < n:
            for end in range(n, start, -1):
                cand = word[start:end] if start == 0 else "##" + word[start:end]
                if cand in self.vocab:
                    pieces.append(cand)
                    start = end
                    break
            else:
                return None
        return pieces

    def tokenize(self, text):
        out = []
        for word in convert_to_unicode(text).split():
            pieces = self._cover(word) if len(word) <= self.max_input_chars_per_word else None
            if pieces is None:
                out.append(self.unk_token)
            else:
                out.extend(pieces)
        return out


class FullTokenizer(object):
    """``FullTokenizer(vocab_file, do_lower_case)``; ``tokenize(text)`` -> list of pieces."""

    def __init__(self, vocab_file, do_lower_case=True):
        self.vocab = load_vocab(vocab_file)
        self.inv_vocab = {v: k for k, v in self.vocab.items()}
        self.basic_tokenizer = BasicTokenizer(do_lower_case=do_lower_case)
        self.wordpiece_tokenizer = WordpieceTokenizer(vocab=self.vocab)

    def tokenize(self, text):
        out = []
        for word in self.basic_tokenizer.tokenize(text):
            out.extend(self.wordpiece_tokenizer.tokenize(word))
        return out

    def convert_tokens_to_ids(self, tokens):
        return [self.vocab[t] for t in tokens]

    def convert_ids_to_tokens(self, ids):
        return [self.inv_vocab[i] for i in ids]
