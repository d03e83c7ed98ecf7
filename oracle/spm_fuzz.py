"""Differential fuzz of build/tokenizer_tool (the C++ tokenizer) against the sentencepiece Python module on the three fixture models:
random mixed-script strings through Encode, their ids and random id sequences through Decode.  Build container only (needs the
module); run as: python oracle/spm_fuzz.py [seed].  Test infrastructure, like everything under oracle/."""
import os, random, subprocess, sys
import sentencepiece as spm
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); GOLD=ROOT+"/tests/golden"; tool=ROOT+"/ppl.llm.serving_amd/build/tokenizer_tool"
random.seed(int(sys.argv[1]) if len(sys.argv)>1 else 0)
alph = list("abcdefghijklmnopqrstuvwxyz  ABCDEFTHE.,!?0123456789\t\n'\"-") + list("éüßñçøåæ中文日本語한국어🙂🚀αβγδ€£—…▁") + ["  ", "   ", " the ", "ing", "tion", "hello", "world", "​", " ", "　", "﻿"]
def rnd():
    n = random.randint(0, 40)
    return "".join(random.choice(alph) for _ in range(n))
bad=0
for model in ["spm_bpe.model","spm_unigram.model","spm_unigram_nofb.model"]:
    sp = spm.SentencePieceProcessor(model_file=os.path.join(GOLD, model))
    texts=[rnd() for _ in range(3000)]
    texts=[t for t in texts if "\n" not in t or True]
    lines=["E "+t.encode().hex() for t in texts]
    out=subprocess.run([tool, os.path.join(GOLD,model)], input="\n".join(lines)+"\n", capture_output=True, text=True, timeout=120)
    rows=out.stdout.split("\n")[1:]
    ids_all=[]
    for t,row in zip(texts,rows):
        got=[int(x) for x in row.split()] if row.strip() else []
        want=sp.encode(t)
        ids_all.append(want)
        if got!=want:
            bad+=1
            if bad<10: print(model,"ENC",repr(t),got,want,[sp.id_to_piece(i) for i in want])
    # decode: random id sequences and the encodings
    V=sp.get_piece_size()
    seqs=ids_all[:1000]+[[random.randrange(V) for _ in range(random.randint(0,12))] for _ in range(3000)]
    lines=["D "+" ".join(map(str,s)) for s in seqs]
    out=subprocess.run([tool, os.path.join(GOLD,model)], input="\n".join(lines)+"\n", capture_output=True, text=True, timeout=120)
    rows=out.stdout.split("\n")[1:]
    for s,row in zip(seqs,rows):
        got=bytes.fromhex(row[1:]).decode("utf-8",errors="surrogateescape")
        want=sp.decode(s)
        if got!=want:
            bad+=1
            if bad<20: print(model,"DEC",s,[sp.id_to_piece(i) for i in s],repr(got),repr(want))
print("bad",bad)
