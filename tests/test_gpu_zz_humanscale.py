"""Last module of the GPU suite: the human-scale PAF comparisons (tests/humanscale.py has the what and why).  The stock binary's three
runs were started by tests/test_gpu_00_humanscale_start.py and have had the rest of the suite's run time to finish."""
import pytest

from humanscale import N_ASM, N_CONTIGS, N_READS_C4, N_READS_NS, N_READS_RR, case as _case

pytestmark = pytest.mark.gpu


def test_north_star_target_defaults(human):
    """10 kbp reads, pi 85, segLength 5000 (the stock binary derives sketchSize 310 for the 3 GB file); one context, then two"""
    _case(human, "northstar", int(0.9 * N_READS_NS), True, sharded=True)


def test_configs2_shape_one_to_one(human):
    """assembly vs reference: --pi 95 -s 10000 -f one-to-one (a sketch of ~40 per 10 kbp: the seed table stays below the 1 GiB where the
    tag layer starts)"""
    _case(human, "configs2", N_ASM, False)


def test_configs2_full_size_one_to_one(human):
    """BASELINE configs[2] at its own size: the whole 3 Gbp assembly (every reference contig diverged by 1 %, with 1-5 Mbp inversions and
    translocations, a third of them on the other strand) against the 3 Gbp reference, --pi 95 -s 10000 -f one-to-one: 300 000 fragments
    chained into a few mappings per contig, bytes equal to the stock binary's"""
    _case(human, "configs2_full", N_CONTIGS, False)


def test_configs4_shape_dense_reference_list(human):
    """--dense --pi 80, 20 kbp reads at 15-20 % error, --rl list of 10 reference files sharing one seqId space (winSketch.hpp:174-214)"""
    _case(human, "configs4", int(0.8 * N_READS_C4), True)


def test_repeat_rich_reference(human):
    """the shape bench.py reports as north_star_target.repeat_rich: 10 kbp reads at pi 85 against a 3 Gbp reference with human-like repeat
    structure (interspersed repeat families over ~45 % of it, satellite arrays, N gaps; bench.make_repeat_rich_reference), defaults.
    Reads that fall into an N gap or a satellite array may stay unmapped in both programs; the bytes must be the same."""
    print("\n[human scale] repeat-rich reference:", human["rr_summary"], flush=True)
    _case(human, "repeat_rich", int(0.7 * N_READS_RR), True)
