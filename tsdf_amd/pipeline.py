"""The per-frame step of BASELINE configs[2] -- bilateral filter, integrate, ray cast + normals, frame after frame as the
reference's kinfu loop integrates them (src/Tools/kinfu.cpp:32-56: one blocking call per frame on host buffers) -- on frames
that live in HBM, with the one piece of a frame that depends on nothing before it -- the bilateral filter of the NEXT frame
-- queued on a second, lower-priority HIP stream while this frame's ray cast runs.

Why a second stream pays at all on one GPU: every kernel of the step fills the chip while it is in full swing, but each ends
with a ramp-down (integrate ~10 us, the two ray kernels ~15 and ~20 us) and the small kernels between them (brick cull, reach
summary, resolve + normals) are chains of memory round trips on a few thousand waves.  A kernel of EQUAL priority beside them
only takes turns with them (measured: no gain); one of LOWER priority gets the slots the main stream cannot
use at that moment.  The filter of frame i + 1 is released when integrate of frame i has finished, so it never runs beside
integrate_kernel (which is bound by memory, not by slots), and it must be done before integrate of frame i + 1 starts: the
main stream waits for its event.  0.342-0.351 -> 0.328-0.333 ms per step on the bench stream (tools/dbg_pipeline.py), 0.324-0.326
with the next frame's brick culling queued behind its filter (next_camera).

Results cannot change: the same kernels run on the same inputs, only earlier."""
import torch

from . import api


class FusionPipeline:
    """step(depth_ptr, camera, vertices_ptr, normals_ptr, next_depth_ptr=None): one frame through filter -> integrate ->
    raycast (+ normals).  `next_depth_ptr`, when given, is the device pointer of the frame the next call will pass: its filter
    is queued now.  All pointers are device pointers to width * height uint16 (depth) / 3 * width * height float32 (maps);
    the depth buffers must stay valid until the frame after them has been processed."""

    def __init__(self, volume, bilateral, raycaster, width, height, overlap=True, release_after_integrate=True, slab_exchange=None):
        """slab_exchange = (hits_mine, hits_all, exchange): the volume is one rank's Z-slab; a step ray casts the slab into
        `hits_mine` ((W*H, 4) float32), calls exchange(stream) -- the frame's all-gather into `hits_all` ((world, W*H, 4)),
        enqueued on or ordered behind `stream` -- and merges the ranks' records into the vertex and normal maps."""
        self.volume, self.bilateral, self.raycaster = volume, bilateral, raycaster
        self.width, self.height = int(width), int(height)
        self.overlap = bool(overlap)
        self.release_after_integrate = bool(release_after_integrate)
        self.slab_exchange = slab_exchange
        # (torch: a lower number is a higher priority; the range on this device is 0 .. -1)
        import os
        # With a slab exchange the two streams get EQUAL priority: ncclAllGather enqueued on a stream of raised priority made the
        # one-rank step 0.58 ms instead of 0.37 (RCCL 2.26.6; torch's collective is not affected), and equal priorities still
        # give 0.366 against 0.381 without the second stream.  TSDF_PIPE_EQUAL_PRIORITY=1 forces that everywhere (diagnostics).
        equal = os.environ.get("TSDF_PIPE_EQUAL_PRIORITY") == "1" or slab_exchange is not None
        self.main = torch.cuda.Stream(priority=0 if equal else -1) if self.overlap else torch.cuda.current_stream()
        self.side = torch.cuda.Stream(priority=0) if self.overlap else None
        volume.set_stream(self.main.cuda_stream)
        n = self.width * self.height
        tiles = ((self.width + 15) // 16) * ((self.height + 15) // 16)
        self._filtered = [torch.empty((n,), dtype=torch.int16, device="cuda") for _ in range(2)]
        self._tile_max = [torch.empty((tiles,), dtype=torch.int16, device="cuda") for _ in range(2)]
        self._integrated = [None, None]     # per buffer: the event after the integrate that last read it
        # (the events are made once and recorded again every other frame: creating one per frame makes the runtime grow its
        # pool of signals now and then, a stall of tens of milliseconds in the middle of a stream)
        self._done_events = [torch.cuda.Event(), torch.cuda.Event()] if self.overlap else None
        self._ready_events = [torch.cuda.Event(), torch.cuda.Event()] if self.overlap else None
        self._ahead = None                  # (depth_ptr, buffer, event on the side stream) of the frame filtered ahead
        self._frames = 0

    def _filter(self, depth_ptr, b, stream):
        self.bilateral.filter_device(depth_ptr, self._filtered[b].data_ptr(), self.width, self.height, bits=16,
                                     stream=stream.cuda_stream, tile_max_ptr=self._tile_max[b].data_ptr())

    def step(self, depth_ptr, camera, vertices_ptr, normals_ptr=None, next_depth_ptr=None, next_camera=None):
        """next_camera (with next_depth_ptr): the next frame's pose is known already (ground-truth trajectories; not when the
        pose comes from tracking against this frame's ray cast) -- its brick culling is queued behind its filter as well."""
        main, W, H = self.main, self.width, self.height
        b = self._frames % 2
        if self._ahead is not None and self._ahead[0] == int(depth_ptr) and self._ahead[1] == b:
            main.wait_event(self._ahead[2])                 # filtered ahead, on the side stream
        else:
            self._filter(depth_ptr, b, main)
        self._ahead = None
        self.volume.integrate_device(self._filtered[b].data_ptr(), W, H, camera, tile_max_ptr=self._tile_max[b].data_ptr())
        if self.overlap:
            done = self._done_events[b]
            done.record(main)
            self._integrated[b] = done
            if next_depth_ptr is not None:
                # released by THIS frame's integrate (never beside integrate_kernel); the other buffer was last read by the
                # previous frame's integrate, which lies before it on the main stream
                gate = done if self.release_after_integrate else self._integrated[1 - b]
                if gate is not None:
                    self.side.wait_event(gate)
                self._filter(next_depth_ptr, 1 - b, self.side)
                if next_camera is not None and self.release_after_integrate:
                    # (the list, the boxes and the plane constants are free once this frame's integrate_kernel is done)
                    self.volume.integrate_prepare_device(self._filtered[1 - b].data_ptr(), W, H, next_camera,
                                                         self._tile_max[1 - b].data_ptr(), self.side.cuda_stream)
                ready = self._ready_events[1 - b]
                ready.record(self.side)
                self._ahead = (int(next_depth_ptr), 1 - b, ready)
        if self.slab_exchange is None:
            self.raycaster.raycast_device(self.volume, camera, vertices_ptr, normals_ptr)
        else:
            hits_mine, hits_all, exchange = self.slab_exchange
            self.raycaster.raycast_slab_device(self.volume, camera, hits_mine.data_ptr())
            exchange(main)
            if normals_ptr is not None:
                api.merge_hits_normals_device(hits_all.data_ptr(), hits_all.shape[0], W, H, vertices_ptr, normals_ptr, main.cuda_stream)
            else:
                api.merge_hits_device(hits_all.data_ptr(), hits_all.shape[0], W, H, vertices_ptr, main.cuda_stream)
        self._frames += 1

    def synchronize(self):
        self.main.synchronize()
        if self.side is not None:
            self.side.synchronize()
