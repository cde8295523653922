"""The live system's control flow around the depth filter, without ROS: rmd::DepthmapNode (src/depthmap_node.cpp:30-182) and
rmd::Publisher (src/publisher.cpp:28-147) as a library, so that a batch job can cycle reference frames exactly like the
node does and its throughput includes the periodic seed_init, TV-L1 and point-cloud steps.

What the reference hands to ROS topics is handed to callbacks here:
    on_depthmap(depth float32 HxW)                      remode/depth        (publisher.cpp:40-52)
    on_pointcloud(points float32 Nx4: x, y, z, I)        remode/pointcloud   (:54-104), the cloud ACCUMULATES over publications
    on_convergence(bgr uint8 HxWx3)                      remode/convergence  (:112-147)
The point cloud is computed on the device from the denoiser's output and the convergence plane
(rmd_hip_seeds_point_cloud); the reference downloads both images and loops on the host.
"""
import numpy as np

from . import api


class State:  # depthmap_node.h:32-36
    TAKE_REFERENCE_FRAME = 0
    UPDATE = 1


class Publisher:
    def __init__(self, depthmap, on_depthmap=None, on_pointcloud=None, on_convergence=None, verbose=False):
        self.depthmap_ = depthmap
        self.on_depthmap, self.on_pointcloud, self.on_convergence = on_depthmap, on_pointcloud, on_convergence
        self.verbose = verbose
        self.pc_ = np.zeros((0, 4), np.float32)  # the reference never clears its cloud: every publication appends (publisher.cpp:83)
        self._pc_store = np.zeros((0, 4), np.float32)  # what pc_ is a view of: grown by doubling, so that appending does not copy the whole cloud every time

    def publishDepthmap(self, depth=None):
        """depth: the map of a collected asynchronous publication (default: the Depthmap's host mirror, as in the reference)"""
        if self.on_depthmap:
            self.on_depthmap(self.depthmap_.getDepthmap() if depth is None else depth)
        if self.verbose:
            print("INFO: publishing depth map")

    def publishPointCloud(self, pts=None):
        """pts: the points of a collected asynchronous publication (default: computed now, synchronously)"""
        if pts is None:
            pts = self.depthmap_.downloadPointCloud(denoised=True)
        if len(pts):
            n0, n1 = len(self.pc_), len(self.pc_) + len(pts)
            if self.pc_.base is not self._pc_store or n1 > len(self._pc_store):  # (a caller may have replaced pc_: start from what it holds)
                store = np.empty((max(2 * n1, 1 << 16), 4), np.float32)
                store[:n0] = self.pc_
                self._pc_store = store
            self._pc_store[n0:n1] = pts
            self.pc_ = self._pc_store[:n1]
        if len(self.pc_):
            if self.on_pointcloud:
                self.on_pointcloud(self.pc_)
            if self.verbose:
                print(f"INFO: publishing pointcloud, {len(self.pc_)} points")

    def publishDepthmapAndPointCloud(self):
        self.publishDepthmap()
        self.publishPointCloud()

    def publishConvergenceMap(self, colored=None):
        """publisher.cpp:112-147.  The library's Depthmap colours the map on the device (rmd_hip_seeds_convergence_bgr8: 3 bytes per pixel
        cross the bus); any other object with rmd::Depthmap's interface (the CPU tests drive the oracle) gets the reference's host loop.
        colored: the map of a collected asynchronous publication."""
        if colored is not None:
            pass
        elif hasattr(self.depthmap_, "convergenceBGR8"):
            colored = self.depthmap_.convergenceBGR8()
        else:
            conv = self.depthmap_.getConvergenceMap()
            colored = np.repeat(self.depthmap_.getReferenceImage()[:, :, None], 3, axis=2)  # CV_GRAY2BGR
            colored[..., 0][conv == api.ConvergenceStates.CONVERGED] = 255
            colored[..., 2][conv == api.ConvergenceStates.DIVERGED] = 255
        if self.on_convergence:
            self.on_convergence(colored)
        if self.verbose:
            print("INFO: publishing convergence map")
        return colored


class DepthmapNode:
    """denseInput() is DepthmapNode::denseInputCallback (depthmap_node.cpp:88-163); the parameters are the node's ROS
    parameters with their defaults (:76-78)."""

    def __init__(self, cam_width, cam_height, cam_fx, cam_fy, cam_cx, cam_cy, ref_compl_perc=10.0, max_dist_from_ref=0.5,
                 publish_conv_every_n=10, patch_side=5, max_extent=100, denoise_lambda=0.5, denoise_iterations=200,
                 on_depthmap=None, on_pointcloud=None, on_convergence=None, verbose=False, distortion=None, depthmap=None, async_publish=False):
        # `depthmap`: an object with rmd::Depthmap's interface to drive instead of the library's (the CPU tests use one)
        # `async_publish`: publication off the update stream (Depthmap.publishAsync).  The reference launches its publisher with std::async and
        # discards the future, whose destructor waits (depthmap_node.cpp:170-172, 179-181): publication blocks the callback.  With this switch
        # the callback only REQUESTS the products (a snapshot of the state, a few microseconds) and goes on; what is published is the same, bit
        # for bit and in the same order, but it reaches the callbacks a few messages later: at the start of a later denseInput() once the
        # device has finished it, or in flush().  Off by default: the default node publishes WHEN the reference does (tests/test_reference_host_sources.py).
        self.async_publish_ = bool(async_publish) and depthmap is None
        self.in_flight_ = []  # kinds ("results" / "convergence") of the publications requested and not yet delivered, oldest first
        self.depthmap_ = depthmap if depthmap is not None else api.Depthmap(cam_width, cam_height, cam_fx, cam_cx, cam_fy, cam_cy,
                                                                            patch_side=patch_side, max_extent=max_extent)
        if distortion is not None:  # remode/cam_k1, cam_k2, cam_r1, cam_r2 (depthmap_node.cpp:66-74)
            self.depthmap_.initUndistortionMap(*distortion)
        self.state_ = State.TAKE_REFERENCE_FRAME
        self.ref_compl_perc_ = np.float32(ref_compl_perc)
        self.max_dist_from_ref_ = np.float32(max_dist_from_ref)
        self.publish_conv_every_n_ = publish_conv_every_n
        self.num_msgs_ = 0
        self.lambda_, self.iterations_ = denoise_lambda, denoise_iterations  # 0.5f, 200 at depthmap_node.cpp:167
        self.publisher_ = Publisher(self.depthmap_, on_depthmap, on_pointcloud, on_convergence, verbose)
        self.verbose = verbose
        self.references_taken = 0
        self.updates_done = 0

    def denseInput(self, img_8uC1, T_world_curr, min_depth, max_depth):
        """One svo_msgs::DenseInput message: 8-bit image, camera pose in the world frame (an api.SE3, 12 floats, or
        (qw, qx, qy, qz, tx, ty, tz)), scene depth range.  Returns the state the node is in afterwards."""
        self.num_msgs_ += 1
        if self.in_flight_:
            self.deliver(wait=False)
        if not isinstance(T_world_curr, api.SE3):
            T_world_curr = api.SE3(*T_world_curr) if len(T_world_curr) == 7 else api.SE3(T_world_curr)
        if self.verbose:
            print(f"DEPTHMAP NODE: received image {img_8uC1.shape[1]}x{img_8uC1.shape[0]}")
            print("T_world_curr:")
            print(T_world_curr)
        if self.state_ == State.TAKE_REFERENCE_FRAME:
            if self.depthmap_.setReferenceImage(img_8uC1, T_world_curr.inv(), min_depth, max_depth):
                self.state_ = State.UPDATE
                self.references_taken += 1
        elif self.state_ == State.UPDATE:
            self.depthmap_.update(img_8uC1, T_world_curr.inv())
            self.updates_done += 1
            perc_conv = np.float32(self.depthmap_.getConvergedPercentage())
            dist_from_ref = np.float32(self.depthmap_.getDistFromRef())
            if self.verbose:
                print(f"INFO: percentage of converged measurements: {perc_conv}%")
            if perc_conv > self.ref_compl_perc_ or dist_from_ref > self.max_dist_from_ref_:
                self.state_ = State.TAKE_REFERENCE_FRAME
                self.denoiseAndPublishResults()
        if self.publish_conv_every_n_ < self.num_msgs_:
            self.publishConvergenceMap()
            self.num_msgs_ = 0
        return self.state_

    def denoiseAndPublishResults(self):  # depthmap_node.cpp:165-173
        if self.async_publish_:
            self._request("results", api.PUBLISH_DEPTH | api.PUBLISH_CLOUD | api.PUBLISH_CONVERGENCE)
            return
        self.depthmap_.downloadDenoisedDepthmap(self.lambda_, self.iterations_)
        self.depthmap_.downloadConvergenceMap()
        self.publisher_.publishDepthmapAndPointCloud()

    def publishConvergenceMap(self):  # :175-182
        if self.async_publish_:
            # (the reference refreshes its int32 host mirror here too, :177, although the colouring below is all that is published: in this mode
            # the mirror -- getConvergenceMap() -- is refreshed by the RESULTS publications only; 1.2 MB per map that nobody reads stay on the device)
            self._request("convergence", api.PUBLISH_CONVERGENCE_BGR)
            return
        self.depthmap_.downloadConvergenceMap()  # (:177: the host mirror is refreshed here whether or not the colouring needs it)
        self.publisher_.publishConvergenceMap()

    # ---- publication off the update stream (async_publish) ----
    MAX_IN_FLIGHT = 3  # RMD_HIP_PUBLISH_SLOTS

    def _request(self, kind, what):
        if len(self.in_flight_) >= self.MAX_IN_FLIGHT:
            self.deliver(wait=True, at_most=1)  # the oldest one's slot is needed
        self.depthmap_.publishAsync(what, self.lambda_, self.iterations_)
        self.in_flight_.append(kind)

    def deliver(self, wait=False, at_most=None):
        """hand the finished publications to the callbacks, oldest first; wait=True blocks until they are.  Returns how many were delivered."""
        n = 0
        while self.in_flight_ and (at_most is None or n < at_most):
            pub = self.depthmap_.collectPublication(wait, views=True)  # points / coloured map: views of the library's pinned buffers ...
            if pub is None:
                break
            kind = self.in_flight_.pop(0)
            try:
                if kind == "results":
                    self.publisher_.publishDepthmap(pub["depth"])
                    self.publisher_.publishPointCloud(pub["points"])  # (appended to the accumulated cloud: its one copy)
                else:
                    self.publisher_.publishConvergenceMap(pub["bgr"])  # (a callback that keeps the image copies it, as with any message buffer)
            finally:
                self.depthmap_.releasePublication()  # ... valid until here
            n += 1
        return n

    def flush(self):
        """everything requested has been published when this returns (a node that shuts down, or a caller that needs the last results now)"""
        return self.deliver(wait=True)
