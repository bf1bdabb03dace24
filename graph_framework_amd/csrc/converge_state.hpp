//------------------------------------------------------------------------------
///  @file converge_state.hpp
///  @brief Device-resident state of one converge loop (workflow.hpp:179-205), shared by the
///  host runtime (gf_hip.cpp) and the decide kernel (reduce.hip).
//------------------------------------------------------------------------------
#ifndef gfhip_converge_state_hpp
#define gfhip_converge_state_hpp

namespace gfhip {

struct converge_state {
    double max_residual;            ///< max of the last pass that ran
    double last, off_last;          ///< the loop's last_max / off_last_max (exact images of the item's type)
    double tolerance;
    unsigned long long iterations;  ///< the loop's `iterations` (post-increment semantics kept)
    unsigned long long limit;       ///< max_iterations
    unsigned int done;              ///< the loop's condition came out false: later passes return at once
    unsigned int passes;            ///< passes that really ran
    unsigned int extra;             ///< batched passes (`<name>_batch`): passes of the last batch that ran BEYOND the loop's last
                                    ///< one — the state is then restored from the undo arrays and the batch redone without them
    unsigned int batch_passes;      ///< passes of the last batch that belong to the loop
};

}  // namespace gfhip

#endif /* gfhip_converge_state_hpp */
