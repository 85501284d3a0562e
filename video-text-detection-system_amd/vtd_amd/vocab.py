"""CRNN vocabulary (reference: app/ml/models/text_recognizer.py:86-91).

95 printable symbols -> ids 1..95, '<blank>' = 0, '<unk>' = 96 (size 97).
"""
import string

# digits, lower, upper, the 32 ASCII punctuation marks in ASCII order, then space
VOCAB_CHARS = string.digits + string.ascii_lowercase + string.ascii_uppercase + string.punctuation + " "
BLANK_ID = 0


def build_vocab():
    table = {ch: i for i, ch in enumerate(VOCAB_CHARS, start=1)}
    table["<blank>"] = BLANK_ID
    table["<unk>"] = len(table)
    return table


def id_to_char_table(vocab):
    """Dense id -> code point table for the device CTC decoder; -1 marks ids that emit
    nothing ('<blank>', '<unk>', and anything outside the table)."""
    n = max(vocab.values()) + 1
    out = [-1] * n
    for k, v in vocab.items():
        if len(k) == 1:
            out[v] = ord(k)
    return out
