"""CPU test: the ctypes mirrors of the C-ABI structs (llmrec_amd/ops.py) have the size and field offsets the C compiler
gives the typedefs of include/llmrec_hip.h - a mismatch would silently corrupt every argument block."""
import ctypes
import os
import re
import subprocess

from llmrec_amd import _lib, ops

PAIRS = {
    "llmrec_spmm_plan_t": ops.SpmmPlanC, "llmrec_spmm_epilogue_t": ops.SpmmEpilogueC, "llmrec_linear_problem_t": ops.LinearProblem,
    "llmrec_wgrad_problem_t": ops.WgradProblem, "llmrec_bpr_problem_t": ops.BprProblem, "llmrec_adamw_tensor_t": ops.AdamwTensor,
    "llmrec_zero_tensor_t": ops.ZeroTensor, "llmrec_wgrad_target_t": ops.WgradTarget,
    "llmrec_fuse_fwd_problem_t": ops.FuseFwdProblem, "llmrec_fuse_bwd_problem_t": ops.FuseBwdProblem,
    "llmrec_wgrad_update_t": ops.WgradUpdate, "llmrec_zero_rows_job_t": ops.ZeroRowsJob,
}


def test_ctypes_structs_match_the_header(tmp_path):
    text = open(_lib.HEADER).read()
    fields = {}
    for name in PAIRS:
        end = re.search(r"\}\s*%s\s*;" % name, text)
        assert end, name
        start = text.rfind("typedef struct", 0, end.start())
        body = text[text.index("{", start) + 1:end.start()]
        body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"\w+", part)[-1])
        fields[name] = names
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % _lib.HEADER, "int main(void) {"]
    for name, fs in fields.items():
        src.append('printf("%s %%zu", sizeof(%s));' % (name, name))
        for f in fs:
            src.append('printf(" %%zu", offsetof(%s, %s));' % (name, f))
        src.append('printf("\\n");')
    src.append("return 0; }")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-o", str(exe), str(c)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for line in out:
        parts = line.split()
        name, size, offs = parts[0], int(parts[1]), [int(x) for x in parts[2:]]
        st = PAIRS[name]
        assert ctypes.sizeof(st) == size, (name, ctypes.sizeof(st), size)
        got = [getattr(st, f[0]).offset for f in st._fields_]
        assert got == offs, (name, got, offs)
        assert [f[0] for f in st._fields_] == fields[name], (name, fields[name])


def test_spmm_shape_policy():
    # launch-bound graphs: 32 nnz per lane group, wide operands in 64-column slices; HBM-bound graphs: 128 / 512
    assert ops.spmm_shape(64, 55_146) == (0, (128, 2048, 2048))
    assert ops.spmm_shape(448, 55_146) == (64, (128, 2048, 2048))
    assert ops.spmm_shape(448, 55_146, whole_row=True) == (0, (32, 512, 512))
    assert ops.spmm_shape(64, 36_000_000) == (0, (512, 16384, 16384))
    assert ops.spmm_shape(128, 1_000_000_000) == (0, (256, 8192, 8192))
    assert ops.spmm_shape(20, 1000)[0] == 0
