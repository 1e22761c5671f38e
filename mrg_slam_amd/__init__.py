"""mrg_slam_amd — MI355X (gfx950) scan-matching front end for mrg_slam: prefilter chain + NDT/GICP align() as
hand-written HIP kernels behind a C ABI (include/mrgfe.h), with the host-side mirror of the reference's
pcl::Registration / pcl::Filter call surface.  See DESIGN.md."""
from ._lib import Context, MrgfeError, build, default_context  # noqa: F401
from .filters import ApproximateVoxelGrid, InformationMatrixCalculator, RadiusOutlierRemoval, StatisticalOutlierRemoval, VoxelGrid, calc_fitness_score, distance_filter, knn, prefilter, prefilter_to_device  # noqa: F401
from .map_cloud import KeyFrameSnapshot, MapCloudGenerator, MapCloudStore, deskew, remove_points_near, transform_cloud  # noqa: F401
from .registration import BatchMatcher, GicpHip, IcpHip, NdtHip, NodeMatcher, PclGicpHip, PclNdtHip, SmallGicpHip, VgicpHip, select_registration_method  # noqa: F401
from .loop_detector import KeyFrame, LoopDetector  # noqa: F401,E402
from .odometry import ScanMatchingOdometry  # noqa: F401,E402
from .prefiltering import PrefilteringComponent  # noqa: F401,E402
