"""vtkcloudpoint_amd -- MI355X-native DBSCAN + centroid + ICP hot path of ZhiHuangHn/vtkCloudPoint.

The compute lives in libvcp.so (hand-written HIP for gfx950 behind the C-ABI of include/vcp.h);
this package is the thin host-side mirror of the reference's class surface.
"""
__version__ = "0.1.0"
