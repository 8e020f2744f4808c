"""CPU oracle for the MoPA hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain numpy / torch-CPU restatement of the algorithms on the
hot path named by BASELINE.json.  It exists so that the
HIP kernels in ``mopa_amd/csrc`` can be checked for parity.  Nothing in the
product package (``mopa_amd``) imports it; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do.

Pinning status
--------------
* 2D branch (``oracle.net2d``), losses (``oracle.losses``), SegIoU and the
  voxeliser (``oracle.voxelize``): PINNED against outputs of the imported
  reference (``/root/reference``, run under sys.modules stubs in the build
  container by ``oracle/gen_golden.py``); the vectors live in ``tests/golden``.
* Pseudo-label update (``oracle.pseudo``): ``refine_pseudo_labels``
  and ``prob_2_entropy`` PINNED by fixture G5; the EMA rule restates the
  published update of the un-pinned pip package ``torch_ema`` (unpinned for
  that one function, anchored on the reference's call sites).
* 3D branch (``oracle.scn3d``): **parity unpinned**.  The arithmetic lives in
  the third-party package ``sparseconvnet`` (facebookresearch/SparseConvNet,
  installed un-pinned from git HEAD by the reference's ``install.sh:1``), whose
  source is neither under ``/root/reference`` nor installed, and the reference
  has no tests or golden vectors at that boundary.  The restatement follows the
  library's published semantics and the reference's
  call sites ``mopa/models/scn_unet.py:9-34`` / unrolled wiring ``:38-219``; it
  is anchored by dense ``conv3d`` / ``conv_transpose3d`` / ``batch_norm``
  equivalences (``tests/test_oracle_scn3d.py``), not by reference outputs.
"""
