"""CPU oracle for the sharkshark-4k upscale hot path.  TEST INFRASTRUCTURE ONLY.

A PyTorch-CPU fp32 restatement of the reference's per-frame super-resolution path
(``/root/reference/src/upscale``), written from the architecture facts in SURVEY.md §8(a) and
checked against the reference's own modules (imported in the build container by
``tests/golden/make_golden.py``; the resulting vectors are committed under ``tests/golden``).

Nothing in the product path (``sharkshark-4k_amd/``) may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, as the checker.

Pinning status
--------------
* FSRCNN, SRVGGNetCompact, BSVD (F=1) and the service glue (``upscale_multi`` /
  ``upscale_single``) are pinned against the imported reference modules (see
  ``tests/golden/MANIFEST.json``; max|delta| recorded there).
* RRDBNet: **parity unpinned**.  The class lives in the third-party package ``basicsr``
  (imported at ``src/upscale/model/realesrgan/factory.py:6``; version not pinned by the
  reference, only "RealESRGAN commit 5ca1078" is, ``README.md:62``) which is absent from this
  image.  ``oracle.nets.rrdbnet`` restates the published BasicSR ``rrdbnet_arch.RRDBNet`` and is
  self-checked by parameter count (16 703 171 for x2 / 16 697 987 for x4) and FLOP count only.
* ``oracle.cv_area`` (the image server's cv2 ``INTER_AREA`` pre / post scale): **parity unpinned** - OpenCV is neither vendored nor
  version-pinned by the reference and is absent from this image; the module restates the published algorithm.
"""
