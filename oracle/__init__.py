"""CPU oracle for the PoseTraj denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``posetraj_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / reported baseline.

It is a plain-PyTorch (CPU, fp32 or fp64) restatement of the algorithm the
reference executes on the path named in ``BASELINE.json``:

* ``sched.py``      - ``utils/scheduling_euler_discrete_karras_fix.py``  (in tree)
* ``cond_embed.py`` - ``models/controlnet_sdv.py:61-116`` and the camera twin
                      ``models/controlnet_sdv_cam_infer.py:61-130``     (in tree)
* ``nets.py``       - ``models/controlnet_sdv.py:201-650`` and
                      ``models/unet_spatio_temporal_condition_controlnet.py``
                      top-level wiring                                  (in tree)
* ``blocks.py``     - the ``diffusers==0.24.0`` block classes those two files
                      instantiate (``requirements.txt:4``).  That package is
                      NOT vendored in the reference, NOT installed here and not
                      fetchable, so this file restates its published algorithm.
* ``loop.py``       - the denoise loop of
                      ``pipeline/pipeline_stable_video_diffusion_controlnet.py:481-583``

Pinning status (see DESIGN.md "Oracle"):
  PINNED against outputs of the reference's own code run in the build
  container (fixtures under ``tests/golden/``, generator
  ``tests/golden/make_golden.py``): sched.py, cond_embed.py, the wiring of
  nets.py (tap order, residual multiplicity, zero-conv / scale order, embedding
  flow) and the loop body of loop.py.
  PARITY UNPINNED: blocks.py.  The reference holds no tests, no golden vectors
  and no copy of diffusers; block internals follow the in-tree restatement
  ``models/modified_svd.py`` where one exists and published diffusers 0.24.0
  behaviour elsewhere.
"""
