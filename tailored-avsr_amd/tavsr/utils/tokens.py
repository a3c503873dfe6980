"""Built-in character inventories (vocabulary size is an input constant of the path: V = 41 English, 37 Spanish;
src/tokenizers/char/{english,spanish}.txt of the reference, the lists its LM recipes spell out)."""

CHAR_ENGLISH = (["<blank>", "<unk>", "'"] + [str(d) for d in range(10)] + ["<space>"]
                + [chr(c) for c in range(ord("A"), ord("Z") + 1)] + ["<sos/eos>"])

CHAR_SPANISH = (["<blank>", "<unk>", "<space>"] + [chr(c) for c in range(ord("A"), ord("Z") + 1)]
                + list("\u00c1\u00c9\u00cd\u00d1\u00d3\u00da\u00dc") + ["<sos/eos>"])      # A-Z, then the accented capitals and N-tilde

BUILTIN = {"char/english": CHAR_ENGLISH, "char/spanish": CHAR_SPANISH}


def load_token_list(spec):
    if isinstance(spec, (list, tuple)):
        return list(spec)
    if spec in BUILTIN:
        return list(BUILTIN[spec])
    with open(spec, encoding="utf-8") as f:
        return [line.rstrip() for line in f]
