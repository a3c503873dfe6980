"""Built-in character inventories (vocabulary size is an input constant of the path: V=41 EN)."""

CHAR_ENGLISH = (["<blank>", "<unk>", "'"] + [str(d) for d in range(10)] + ["<space>"]
                + [chr(c) for c in range(ord("A"), ord("Z") + 1)] + ["<sos/eos>"])

BUILTIN = {"char/english": CHAR_ENGLISH}


def load_token_list(spec):
    if isinstance(spec, (list, tuple)):
        return list(spec)
    if spec in BUILTIN:
        return list(BUILTIN[spec])
    with open(spec, encoding="utf-8") as f:
        return [line.rstrip() for line in f]
