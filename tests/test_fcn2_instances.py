"""CPU: the kernel instances lm_fcn2.hip's dispatcher holds (the LM_G2_TRY lists of lm_g2_launch) and lecturemath_amd/fcn2.py: have_instance --
the table the recipes consult before they ask for a variant -- name the same set.  A variant only one of them knows is either a launch
that fails at run time or a kernel nobody can reach."""
import itertools
import os
import re

from lecturemath_amd import fcn2

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lecturemath_amd", "csrc", "lm_fcn2.hip")
EPI = {"LM_G2_EPI_PO": fcn2.EPI_PO, "LM_G2_EPI_T": fcn2.EPI_T, "LM_G2_EPI_TC": fcn2.EPI_TC, "LM_G2_EPI_V": fcn2.EPI_V}


def dispatcher_instances():
    text = open(SRC).read()
    body = text[text.index("static int lm_g2_launch(const LmF2Layer& l"):text.index("#undef LM_G2_TRY\n")]
    m77 = re.search(r"#define LM_G2_TRY77\(N, L\) \\\n(.*?)\n\s*LM_G2_TRY77\(", body, re.S)
    assert m77, "LM_G2_TRY77 not found"
    for n, l in re.findall(r"LM_G2_TRY77\((\d), (\d)\)", body[m77.end() - 12:]):
        body += "\n" + m77.group(1).replace("N, L)", "%s, %s)" % (n, l))
    out = set()
    for kh, kw, t, e, n, l in re.findall(r"LM_G2_TRY_MT4\((\d), (\d), (\d), (\w+), (\d), (\d)\)", body):
        for m in (1, 2, 3, 4):
            out.add((int(kh), int(kw), int(t), m, EPI[e], int(n), int(l)))
    for kh, kw, t, m, e, n, l in re.findall(r"LM_G2_TRY\((\d), (\d), (\d), (\d), (\w+), (\d), (\d)\)", body):
        out.add((int(kh), int(kw), int(t), int(m), EPI[e], int(n), int(l)))
    return out


def test_dispatcher_and_have_instance_agree():
    have = dispatcher_instances()
    assert len(have) > 60
    grid = set()
    for (kh, kw), epi in (((3, 3), fcn2.EPI_PO), ((1, 1), fcn2.EPI_TC), ((1, 7), fcn2.EPI_T), ((1, 7), fcn2.EPI_V), ((7, 7), fcn2.EPI_PO)):
        for terms, mt, nc, loader in itertools.product((1, 2, 3, 4), (1, 2, 3, 4), (1, 2), (0, 1)):
            if fcn2.have_instance(kh, kw, terms, mt, epi, nc, loader):
                grid.add((kh, kw, terms, mt, epi, nc, loader))
    assert grid == have, ("only in fcn2.have_instance: %s; only in lm_g2_launch: %s" % (sorted(grid - have), sorted(have - grid)))
    # the merged transposed convolution (EPI_TC2) launches the EPI_TC template with four channel tiles
    assert fcn2.have_instance(1, 1, 1, 4, fcn2.EPI_TC2) and (1, 1, 1, 4, fcn2.EPI_TC, 1, 0) in have
