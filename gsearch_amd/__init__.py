"""gsearch_amd — MI355X-native (gfx950) sketch-and-query hot path of gsearch behind a C ABI.

See DESIGN.md / SPEC.md / INTEGRATION.md. Importing the package does not need a GPU; the first call
that touches the device does, and fails loudly otherwise (no CPU fallback).
"""
from ._lib import ALGO, DATA, GsError, SO_PATH, SYMBOLS, load  # noqa: F401
from .api import (Comm, Context, DistHamming, Hnsw, HyperLogLogSketch, Neighbour, OptDensHashSketch, ProbHash3aSketch,  # noqa: F401
                  RevOptDensHashSketch, ReqAnswer, SeqSketcherParams, SuperHash2Sketch, SuperHashSketch, ani, bindash_distance, bindash_sketch_params,
                  default_context, fasta_scan, filter_aa_records, is_fasta_file, list_fasta_files, gunzip_batch, read_fasta_file, pack_dna_records, sketch_fasta_files, sketcher_for, topk_block_bytes, topk_merge_dev, topk_pack, topk_unpack)
