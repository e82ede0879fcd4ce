//! Drop-in for src/clustering.rs of dkohlsdorf/audio_pattern_discovery: Merge, ClusteringOperation and the two associated
//! functions main.rs calls (clustering.rs:8-25, 40-76, 81-110), bodies on the MI355X.  UNCOMPILED (no Rust toolchain here).
//!
//! NOT carried over: `pub fn merge(&mut self) -> ClusteringOperation` (clustering.rs:175-209) and the struct's state it steps
//! (`parents`, `distances`, `n_instances`: clustering.rs:27-33).  It is `pub` in the reference but called from exactly one place,
//! the loop inside `clustering` (clustering.rs:104-107); on the GPU that loop runs device-side (three launches per merge, replayed as
//! a hipGraph) and a host-visible single step would mean a synchronisation and a state download per merge.  A caller that
//! wants the dendrogram one merge at a time reads the returned `Vec<ClusteringOperation>` in order: entry t IS what the t-th `merge()`
//! call returns.
use crate::apd_sys::*;
use std::collections::HashSet;

/// clustering.rs:8-13
#[derive(Debug)]
pub enum Merge {
    Sequence2Sequence,
    Sequence2Cluster,
    Cluster2Sequence,
    Cluster2Cluster,
}

/// clustering.rs:19-25 (`distance` private, as there)
#[derive(Debug)]
pub struct ClusteringOperation {
    pub merge_i: usize,
    pub merge_j: usize,
    pub into: usize,
    distance: f32,
    pub operation: Merge,
}

impl ClusteringOperation {
    fn to_c(&self) -> apd_cluster_op {
        apd_cluster_op { merge_i: self.merge_i as u32, merge_j: self.merge_j as u32, into: self.into as u32, distance: self.distance,
                         operation: match self.operation { Merge::Sequence2Sequence => 0, Merge::Sequence2Cluster => 1,
                                                           Merge::Cluster2Sequence => 2, Merge::Cluster2Cluster => 3 } }
    }
}

/// The reference keeps parents / distances / counters here (clustering.rs:31-37); on the GPU that state lives in HBM for the
/// duration of `clustering`, so the type only carries the associated functions.
pub struct AgglomerativeClustering;

fn with_context<T>(f: impl FnOnce(*mut apd_context) -> T) -> T {
    let device = std::env::var("APD_DEVICES").ok().and_then(|s| s.split(',').next().and_then(|t| t.trim().parse().ok())).unwrap_or(0);
    let mut ctx = std::ptr::null_mut();
    unsafe { check(apd_create(device, &mut ctx)); }
    let out = f(ctx);
    unsafe { apd_destroy(ctx); }
    out
}

impl AgglomerativeClustering {
    /// clustering.rs:81-110: percentile threshold over all n*n values, then average-linkage merges until a linkage reaches
    /// it (that merge is still emitted).  Exact linkage ties take the lowest (merge_i, merge_j); the reference iterates a
    /// HashSet there (clustering.rs:180-181), i.e. no order of its own.
    pub fn clustering(distances: Vec<f32>, n_instances: usize, perc: f32) -> (Vec<ClusteringOperation>, HashSet<usize>) {
        let mut ops = vec![apd_cluster_op::default(); n_instances.max(1)];
        let mut roots = vec![0u32; n_instances.max(1)];
        let (mut n_ops, mut n_roots, mut threshold) = (0u32, 0u32, 0f32);
        with_context(|ctx| unsafe {
            check(apd_clustering(ctx, distances.as_ptr(), 0, n_instances as u32, perc, ops.as_mut_ptr(), &mut n_ops,
                                 roots.as_mut_ptr(), &mut n_roots, &mut threshold));
        });
        println!("Clustering with {}", threshold);                                  // clustering.rs:102
        let ops = ops[..n_ops as usize].iter().map(|o| ClusteringOperation {
            merge_i: o.merge_i as usize, merge_j: o.merge_j as usize, into: o.into as usize, distance: o.distance,
            operation: match o.operation { 0 => Merge::Sequence2Sequence, 1 => Merge::Sequence2Cluster,
                                           2 => Merge::Cluster2Sequence, _ => Merge::Cluster2Cluster },
        }).collect();
        (ops, roots[..n_roots as usize].iter().map(|r| *r as usize).collect())
    }

    /// clustering.rs:40-76: leaf lists per root; roots that were never merged are skipped ("Cluster not found").
    pub fn cluster_sets(operations: &[ClusteringOperation], cluster_ids: &HashSet<usize>, n_instances: usize) -> Vec<Vec<usize>> {
        let ops: Vec<apd_cluster_op> = operations.iter().map(|o| o.to_c()).collect();
        let mut roots: Vec<u32> = cluster_ids.iter().map(|r| *r as u32).collect();
        roots.sort();                                                               // the reference walks the HashSet: unspecified order
        let mut members = vec![0u32; n_instances + ops.len() + 2];
        let mut set_off = vec![0u32; roots.len() + 1];
        let mut n_sets = 0u32;
        unsafe {
            check(apd_cluster_sets(ops.as_ptr(), ops.len() as u32, roots.as_ptr(), roots.len() as u32, n_instances as u32,
                                   members.as_mut_ptr(), set_off.as_mut_ptr(), &mut n_sets));
        }
        (0..n_sets as usize).map(|s| members[set_off[s] as usize..set_off[s + 1] as usize].iter().map(|v| *v as usize).collect()).collect()
    }
}
