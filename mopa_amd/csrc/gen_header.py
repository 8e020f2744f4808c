#!/usr/bin/env python3
"""Regenerate include/mopa_hip.h from the MOPA_API definitions in csrc/*.hip (prototypes only; the section
comments -- which reference interface each group replaces -- are maintained here)."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "..", "include", "mopa_hip.h")

SECTIONS = [
    ("hash3d.hip", """Voxel hash, active sets and rule tables (integer; bit-exact with oracle/scn3d.py::Geometry).
 * Replaces sparseconvnet's host-side Metadata (hash map + rulebooks) reached through
 *   mopa/models/scn_unet.py:26  scn.InputLayer(3, full_scale, mode=4)        -> mopa_voxel_hash_build, mopa_points_csr
 *   mopa/models/scn_unet.py:27  scn.SubmanifoldConvolution(3, cin, m, 3)      -> mopa_rulebook_subm
 *   mopa/models/scn_unet.py:28  scn.UNet(...) Convolution / Deconvolution 2,2  -> mopa_coarsen_build, mopa_rulebook_updown
 * Input contract: coords (N,4) int64 [x,y,z,batch] (mopa/data/collate.py:183-185), 0 <= x,y,z < 4096."""),
    ("spconv.hip", """Sparse convolution on rule tables nbr[K][num_out] (int32, -1 = no rule): out[i] = sum_o in[nbr[o][i]] @ W[o].
 * Replaces the gather-GEMM-scatter kernels of sparseconvnet behind
 *   mopa/models/scn_unet.py:27-28 (SubmanifoldConvolution K=27; scn.UNet Convolution / Deconvolution K=8)
 * forward / backward-data: mopa_spconv_fwd (backward-data = same call on the reversed table with transposed weights);
 * backward-weight: mopa_spconv_bwd_weight."""),
    ("sprun.hip", """The same convolutions, offset-major: run-major rulebook (the rules of one filter offset are one contiguous run of slots),
 * per-offset gather-GEMM with the weight slice resident in LDS, products into a partial slab, one ordered per-row sum -- for the
 * launches the matrix pipe bounds (same reference call sites as spconv.hip; SURVEY A.8 "gather-GEMM-scatter")."""),
    ("exec2d.hip", """Command-list executor: a recorded sequence of this library's own entry points (plus event record / wait between two
 * streams) replayed in ONE call -- the 2D branch's forward / backward pass (mopa/models/resnet34_unet.py:131-191 behind
 * mopa/models/xmuda_arch.py:49-79) without one interpreter round trip per launch; recorder: mopa_amd/dense2d.py::Graph2D."""),
    ("scn_exec.hip", """Native executor of the 3D branch: the whole UNetSCN forward / backward (scn.Sequential of mopa/models/scn_unet.py:25-30
 * with scn.UNet unrolled + the linear heads of mopa/models/xmuda_arch.py:114-126) as ONE call each over host-side tables
 * (layer program, parameter pointers, rule tables, buffer addresses); table layouts: scn_exec.hip / mopa_amd/sparse3d.py."""),
    ("rows.hip", """Row-wise ops over active rows [rows][C] (row stride ld): BatchNorm(+residual)(+Leaky)ReLU, InputLayer, OutputLayer+heads.
 * Replaces  scn.BatchNormReLU / BatchNormLeakyReLU (mopa/models/scn_unet.py:28-29; eps 1e-4),
 *           torch.nn.BatchNorm2d + ReLU (+ residual add of torchvision BasicBlock) of mopa/models/resnet34_unet.py:95,115-129,
 *           scn.InputLayer mode 4 / scn.OutputLayer (mopa/models/scn_unet.py:26,30),
 *           nn.Linear heads on per-point features (mopa/models/xmuda_arch.py:73-77,116,124) and the integer point
 *           gather x[i][idx[:,0], idx[:,1]] (mopa/models/xmuda_arch.py:62-65) -> mopa_output_layer_heads_{fwd,bwd}."""),
    ("conv2d.hip", """Dense 2D convolution (NHWC fp32 implicit GEMM; f32-operand MFMA by default, the fp32 vector pipe with MOPA_CONV2D_MFMA=0): Conv2d fwd / backward-data / backward-weight and
 * ConvTranspose2d(k2,s2) through one index map (25 x int32 'geom', see conv2d.hip::ConvGeom).
 * Replaces the cuDNN calls behind mopa/models/resnet34_unet.py:93 (conv1 7x7), :97-101 (layer1-4), :104-110,:115-129 (decoder)."""),
    ("wino2d.hip", """Winograd F(2x2,3x3) transforms for the stride-1 3x3 convolutions with >= 128 channels (same reference call sites as
 * conv2d.hip); the 16 point-wise GEMMs run through mopa_conv2d_igemm_batched."""),
    ("wino4c9.hip", """The same one-kernel F(4x4) convolution as mopa_wino4_conv (wino2d.hip), nine transform points per wave on v_mfma_f32_32x32x2_f32:
 * half the weight bytes per tile and a quarter of the weight loads per matrix instruction (same reference call sites)."""),
    ("wino4wg.hip", """Weight gradient of the stride-1 3x3 convolutions through Winograd F(4x4,3x3) in ONE kernel (x and dY in, dW out: neither
 * transformed operand reaches HBM) -- torch autograd's conv2d weight gradient of layer1's BasicBlocks and the decoder's 3x3
 * convolutions (mopa/models/resnet34_unet.py:97,104-110,176-182); mopa_wino4_bwd_weight (conv2d.hip) is the two-operand form."""),
    ("ops2d.hip", """MaxPool 3x3/s2/p1 (resnet34_unet.py:148), Dropout p=0.4 (:154,:159), full-image linear head
 * (mopa/models/xmuda_arch.py:58-60), column sums for conv-bias gradients (decoder convs have bias=True, :104-110)."""),
    ("losses.hip", """Losses: cross-modal KL (mopa/train/train_xmuda_mopa.py:389-398,440-445), weighted CE with ignore_index -100
 * (:354-363,456-465,563-567), softmax over classes (:473) and mask_cons_loss (mopa/common/utils/loss.py:241-283).
 * Loss scalars, normalisers and upstream gradients are device pointers (no host sync)."""),
    ("pseudo.hip", """Pseudo-label update of the MoPA phase on the device (SURVEY 8f-3): EMA teacher (torch_ema update rule,
 * mopa/train/train_xmuda_mopa.py:221-226,587-591), entropy-weighted 2D/3D fusion (:282-291 with mopa/models/losses.py:10-19)
 * and the per-class median refinement (mopa/data/utils/refine_pseudo_labels.py:5-22), without host round trips."""),
    ("vgi.hip", """Valid Ground-based Insertion on the device (SURVEY 8f-4): overlap test of the object's box against the scan's
 * occupancy (mopa/data/mixmatch_ss.py:215-331 check_overlap), centre filters (:139-160), ground-cell lookup and road height
 * (:355-455 obj_on_road), range-image occlusion culling (mopa/data/utils/augmentation_3d.py:81-111,161-290) and the float64
 * voxeliser of post_process (mixmatch_ss.py:458-559)."""),
    ("optim.hip", """Adam on one flat fp32 buffer == torch.optim.Adam as built by mopa/common/solver/build.py:7-21 (yaml BASE_LR 1e-3)."""),
]

HEAD = """/* libmopa_hip.so -- C ABI of the MI355X-native MoPA hot path (gfx950 only).
 *
 * GENERATED by mopa_amd/csrc/gen_header.py from the MOPA_API definitions; do not edit by hand.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; `stream` is a hipStream_t;
 *   - plain pointers and sizes only (no torch types); no allocation, no global state, re-entrant per stream;
 *   - scratch memory is passed in (`ws`, `ws_bytes`); the matching *_workspace_bytes() query gives the size;
 *   - return value: 0 = ok, -1 = bad argument / unsupported shape, -2 = workspace too small, -3 = launch failed;
 *   - features are fp32 row-major [rows][ld] (ld >= C lets a layer address a channel slice of a wider buffer).
 * The reference has no FFI of its own (it is pure Python on torch + sparseconvnet); each group below names the
 * reference call sites whose arithmetic it replaces.  Binding example: INTEGRATION.md (ctypes), mopa_amd/_lib.py.
 */
#ifndef MOPA_HIP_H
#define MOPA_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
"""


def protos(path):
    src = open(path).read()
    out = []
    for m in re.finditer(r"MOPA_API\s+([^{;]+?)\s*\{", src, re.S):
        sig = " ".join(m.group(1).split())
        out.append(sig + ";")
    return out


def gen_exec_table():
    """csrc/exec_table.inc: one `case` per launching entry point (returns int, last parameter `void* stream`) for the command-list
    executor of exec2d.hip -- arguments arrive as 64-bit slots (integers and pointers as int64, float / double as the bit pattern of
    a double) and are cast back to the prototype's types.  Ids are positions in the sorted name list (mopa_exec_fn_id)."""
    protos_all = []
    for fname, _ in SECTIONS:
        if fname == "exec2d.hip":
            continue
        protos_all += protos(os.path.join(HERE, fname))
    fns = []
    for sig in protos_all:
        m = re.match(r"(\w[\w\s\*]*?)\s*(mopa_\w+)\((.*)\);$", sig)
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        if ret != "int" or not params.rstrip().endswith("void* stream"):
            continue
        types = []
        for prm in params.split(","):
            prm = re.sub(r"/\*.*?\*/", "", prm).strip()
            t = prm[:prm.rindex(" ")].strip() if " " in prm else prm
            if "*" in prm:
                t = prm[:prm.rindex("*") + 1].strip()
            types.append(t)
        fns.append((name, types))
    fns.sort()
    lines = ["// GENERATED by gen_header.py::gen_exec_table from the MOPA_API prototypes; do not edit by hand.",
             "static const char* const EXEC_NAMES[] = {"]
    lines += [f'  "{n}",' for n, _ in fns]
    lines += ["};", f"static const int EXEC_N = {len(fns)};",
              "static int exec_dispatch(int id, const int64_t* a, int nargs) {", "  switch (id) {"]
    for i, (n, types) in enumerate(fns):
        args = []
        for j, t in enumerate(types):
            if "*" in t:
                args.append(f"({t})(uintptr_t)a[{j}]")
            elif t in ("float", "double"):
                args.append(f"({t})slot_f(a[{j}])")
            else:
                args.append(f"({t})a[{j}]")
        lines.append(f"    case {i}: return nargs == {len(types)} ? {n}({', '.join(args)}) : MOPA_ERR_ARG;")
    lines += ["    default: return MOPA_ERR_ARG;", "  }", "}", ""]
    with open(os.path.join(HERE, "exec_table.inc"), "w") as f:
        f.write("\n".join(lines))
    print("wrote exec_table.inc", len(fns), "entry points")


def gen_host_args():
    """mopa_amd/_host_args.py: for every entry point, the positions and names of its HOST-pointer parameters (names ending in
    `_host`).  The command-list recorder (mopa_amd/_lib.py::CommandList) copies those it knows the size of into its own blob and
    refuses to record an entry point with any other one -- a recorded raw host address would dangle at replay."""
    table = {}
    for fname, _ in SECTIONS:
        for sig in protos(os.path.join(HERE, fname)):
            m = re.match(r"(\w[\w\s\*]*?)\s*(mopa_\w+)\((.*)\);$", sig)
            name, params = m.group(2), m.group(3)
            host = {}
            for j, prm in enumerate(params.split(",")):
                prm = re.sub(r"/\*.*?\*/", "", prm).strip()
                pname = re.split(r"[\s\*]+", prm)[-1]
                if pname.endswith("_host") and "*" in prm:
                    host[j] = pname
            if host:
                table[name] = host
    out = os.path.join(HERE, "..", "_host_args.py")
    with open(out, "w") as f:
        f.write('"""GENERATED by csrc/gen_header.py::gen_host_args from the MOPA_API prototypes; do not edit by hand.\n'
                'entry point -> {argument position: name} of its host-pointer parameters (names ending in _host)."""\n')
        f.write("HOST_PARAMS = {\n")
        for n in sorted(table):
            f.write(f"    {n!r}: {table[n]!r},\n")
        f.write("}\n")
    print("wrote _host_args.py", len(table), "entry points with host pointers")


def main():
    gen_exec_table()
    gen_host_args()
    parts = [HEAD]
    for fname, doc in SECTIONS:
        parts.append(f"\n/* ---- {fname}\n * {doc}\n */")
        parts.extend(protos(os.path.join(HERE, fname)))
    parts.append("\n#ifdef __cplusplus\n}\n#endif\n#endif /* MOPA_HIP_H */\n")
    with open(OUT, "w") as f:
        f.write("\n".join(parts))
    print("wrote", os.path.normpath(OUT), sum(len(protos(os.path.join(HERE, f))) for f, _ in SECTIONS), "prototypes")


if __name__ == "__main__":
    main()
