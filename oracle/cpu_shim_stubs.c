/* oracle/cpu_shim_stubs.c -- TEST INFRASTRUCTURE (see cpu_shim.cpp).  lmono_host.o holds every class of the host mirror; the classes
 * outside the Estimator path (scanRegistration, laserOdometry, laserMapping, MapBuilder, PoseGraph) reference C-ABI entry points the CPU
 * baseline binary never calls.  They resolve to these refusals: no prototype on purpose (this file does not include the header), every one
 * answers LMONO_ENODEV / NULL. */
#define REFUSE_INT(name) long name(void) { return -2; }
#define REFUSE_PTR(name) void *name(void) { return 0; }
#define REFUSE_VOID(name) void name(void) { }
REFUSE_INT(lmono_associate_to_map) REFUSE_PTR(lmono_batch_create) REFUSE_VOID(lmono_batch_destroy) REFUSE_INT(lmono_batch_get_cloud)
REFUSE_INT(lmono_map_builder_clear) REFUSE_INT(lmono_map_builder_cloud) REFUSE_PTR(lmono_map_builder_create) REFUSE_INT(lmono_map_builder_depth)
REFUSE_VOID(lmono_map_builder_destroy) REFUSE_INT(lmono_map_builder_map) REFUSE_PTR(lmono_mapper_create) REFUSE_INT(lmono_mapper_cube)
REFUSE_VOID(lmono_mapper_destroy) REFUSE_INT(lmono_mapper_process) REFUSE_INT(lmono_odom_batch) REFUSE_INT(lmono_odom_step)
REFUSE_PTR(lmono_odom_stream_create) REFUSE_VOID(lmono_odom_stream_destroy) REFUSE_INT(lmono_odom_stream_scan) REFUSE_PTR(lmono_pose_graph_create)
REFUSE_VOID(lmono_pose_graph_destroy) REFUSE_INT(lmono_pose_graph_optimize) REFUSE_INT(lmono_pose_graph_result) REFUSE_INT(lmono_scanreg_batch)
REFUSE_INT(lmono_scanreg_batch_h)
