"""JSON state files of a gsearch database directory (the serde side only):

* ``parameters.json``       — ProcessingParams{hnsw: HnswParams, sketch: SeqSketcherParams, block_flag}  (src/utils/parameters.rs:33-41,139-221)
* ``processing_state.json`` — ProcessingState{nb_seq, nb_file, elapsed_t}                                (src/utils/files.rs:22-111)
* ``seqdict.json``          — SeqDict: ItemDict{id: Id{path, fasta_id}, len} objects written back to back
                              with NO separator (src/utils/idsketch.rs:18-23,130-135,164-253)

Written exactly as serde_json's compact writer does (field order = declaration order, no spaces), so the files are
interchangeable with the reference's. Known-answer: the parameters.json of the README's sample database is 180 bytes
(README.md:164-168) — tests/test_state_json.py. The hnsw_rs graph/data dump formats are NOT reproduced (their layout
lives in an un-vendored crate); the index has its own dump (gs_index_save / gs_index_load).
"""
import json
import os

SKETCH_ALGO = {0: "PROB3A", 1: "SUPER", 2: "SUPER2", 3: "HLL", 4: "OPTDENS", 5: "REVOPTDENS"}
SKETCH_ALGO_INV = {v: k for k, v in SKETCH_ALGO.items()}
DATA_TYPE = {0: "DNA", 1: "AA"}
DATA_TYPE_INV = {v: k for k, v in DATA_TYPE.items()}


def _dumps(obj):
    return json.dumps(obj, separators=(",", ":"), ensure_ascii=False)


def _f(x):
    """serde_json prints f64/f32 with the shortest round-trip representation and always a fractional part"""
    s = repr(float(x))
    return float(s)


class HnswParams:
    """src/utils/parameters.rs:33-41 ; HnswParams::new(capacity, ef, max_nb_conn, scale_modification)"""

    def __init__(self, capacity, ef, max_nb_conn, scale_modification):
        self.capacity, self.ef, self.max_nb_conn, self.scale_modification = int(capacity), int(ef), int(max_nb_conn), float(scale_modification)

    def get_ef(self):
        return self.ef

    def get_max_nb_connection(self):
        return self.max_nb_conn

    def get_scale_modification(self):
        return self.scale_modification

    def to_obj(self):
        return {"capacity": self.capacity, "ef": self.ef, "max_nb_conn": self.max_nb_conn, "scale_modification": _f(self.scale_modification)}


class ProcessingParams:
    """src/utils/parameters.rs:139-221 — sketch = (kmer_size, sketch_size, algo, data_t) of kmerutils::sketcharg::SeqSketcherParams"""

    def __init__(self, hnsw, kmer_size, sketch_size, algo, data_t=0, block_flag=False):
        self.hnsw, self.kmer_size, self.sketch_size = hnsw, int(kmer_size), int(sketch_size)
        self.algo = algo if isinstance(algo, int) else SKETCH_ALGO_INV[algo.upper()] if algo.upper() in SKETCH_ALGO_INV else {"prob": 0, "super": 1, "super2": 2, "hll": 3, "optdens": 4, "revoptdens": 5}[algo]
        self.data_t = data_t if isinstance(data_t, int) else {"dna": 0, "aa": 1, "DNA": 0, "AA": 1}[data_t]
        self.block_flag = bool(block_flag)

    def get_hnsw_params(self):
        return self.hnsw

    def get_kmer_size(self):
        return self.kmer_size

    def get_block_flag(self):
        return self.block_flag

    def to_json(self):
        return _dumps({"hnsw": self.hnsw.to_obj(),
                       "sketch": {"kmer_size": self.kmer_size, "sketch_size": self.sketch_size, "algo": SKETCH_ALGO[self.algo], "data_t": DATA_TYPE[self.data_t]},
                       "block_flag": self.block_flag})

    def dump_json(self, dirpath):
        with open(os.path.join(dirpath, "parameters.json"), "w") as f:
            f.write(self.to_json())

    @classmethod
    def reload_json(cls, dirpath):
        o = json.load(open(os.path.join(dirpath, "parameters.json")))
        h, s = o["hnsw"], o["sketch"]
        return cls(HnswParams(h["capacity"], h["ef"], h["max_nb_conn"], h["scale_modification"]), s["kmer_size"], s["sketch_size"],
                   SKETCH_ALGO_INV[s["algo"]], DATA_TYPE_INV[s["data_t"]], o["block_flag"])


class ProcessingState:
    """src/utils/files.rs:22-111"""

    def __init__(self, nb_seq=0, nb_file=0, elapsed_t=0.0):
        self.nb_seq, self.nb_file, self.elapsed_t = int(nb_seq), int(nb_file), float(elapsed_t)

    def to_json(self):
        return _dumps({"nb_seq": self.nb_seq, "nb_file": self.nb_file, "elapsed_t": _f(self.elapsed_t)})

    def dump_json(self, dirpath):
        with open(os.path.join(dirpath, "processing_state.json"), "w") as f:
            f.write(self.to_json())

    @classmethod
    def reload_json(cls, dirpath):
        o = json.load(open(os.path.join(dirpath, "processing_state.json")))
        return cls(o["nb_seq"], o["nb_file"], o["elapsed_t"])


class SeqDict:
    """src/utils/idsketch.rs:155-253 — rank in the file = data id used in the Hnsw (idsketch.rs:14-16)"""

    def __init__(self, items=None):
        self.items = list(items or [])          # (path, fasta_id, len)

    def append(self, path, fasta_id, length):
        self.items.append((str(path), str(fasta_id), int(length)))

    def get_nb_entries(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def dump(self, filename):
        with open(filename, "w") as f:
            for path, fid, length in self.items:                         # objects back to back, no separator (idsketch.rs:186-189)
                f.write(_dumps({"id": {"path": path, "fasta_id": fid}, "len": length}))

    @classmethod
    def reload_json(cls, filename):
        text = open(filename).read()
        dec, pos, items = json.JSONDecoder(), 0, []
        while pos < len(text):
            while pos < len(text) and text[pos].isspace():
                pos += 1
            if pos >= len(text):
                break
            o, pos = dec.raw_decode(text, pos)                           # streaming reload (idsketch.rs:225)
            items.append((o["id"]["path"], o["id"]["fasta_id"], o["len"]))
        return cls(items)
