"""CPU oracle for the PASTA-GAN++ synthesis hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker / the timed
CPU baseline.  The product (``pasta-gan-plusplus_amd/``) never imports it and
fails loudly when its HIP extension is missing.

What is here
------------
* ``ops_ref.py``      -- pure-torch/NumPy CPU restatement of the reference's
                         operator semantics (upfirdn2d, bias_act, conv2d_resample,
                         fma, modulated_conv2d), each function citing the
                         reference file:line it follows.
* ``network_ref.py``  -- CPU restatement of the synthesis stack
                         (SynthesisNetworkFull_v18 and its blocks) on top of
                         ``ops_ref``; includes the build's own ``SynthesisLayer``
                         (missing from the reference tree).
* ``c/pg_oracle.c``   -- scalar C restatement of the two native plugin kernels
                         (upfirdn2d.cu:29-92 "large" kernel, bias_act.cu:23-147)
                         and a direct-loop conv2d, built by ``oracle/build.py``
                         into ``oracle/_build/libpg_oracle.so``.

Pinning
-------
The reference ships no tests and no golden vectors, so
the oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF: the script
``tests/golden/make_golden.py`` imports ``/root/reference`` (possible only in
the build container), runs the reference's Python-fallback ops and network
classes on seeded inputs, and writes ``tests/golden/*.npz``.  ``tests/
test_oracle_golden.py`` checks every oracle function against those vectors.
"""
