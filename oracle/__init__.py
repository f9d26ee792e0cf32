"""CPU oracle for the colour-transfer hot path.

TEST INFRASTRUCTURE ONLY.  This package is a float64 numpy restatement of the
reference algorithms (egorchistov/color-transfer @ 2024_10_08):

* ``oracle.lab``        -- sRGB<->CIE-Lab exactly as scikit-image 0.18.3
                           ``skimage/color/colorconv.py`` computes it (third
                           party, unpinned in the reference's requirements.txt:2;
                           call sites methods/linear.py:5,25,26,40).
* ``oracle.linear``     -- methods/linear.py:8-124 (Reinhard, Xiao, MK).
* ``oracle.iterative``  -- methods/iterative.py:8-59 (Pitie IDT), with numpy's
                           ``histogram`` / ``interp`` rules restated explicitly.
* ``oracle.pasm`` / ``oracle.dcmcs3di`` -- pasmnet/*.py, methods/dcmcs3di.py:29-66
                           forward pass in plain torch-CPU float64.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the *checker*.  The product path
(``color-transfer_amd/``) never imports anything from here and fails loudly when
the HIP library is missing.

Parity pin: every function here is checked against golden vectors produced by
importing the real reference in the build container
(``tests/golden/make_golden_*.py`` -> ``tests/golden/*.npz``; see
``tests/test_oracle_golden.py``).
"""
