// dx-dap: the partitioner CLI behind scripts/run_dap.sh of the reference.
//
//   dx-dap <dataset> <partitions> <base path> <instances.nt> <schema path>
//
// Same positional arguments as org.ics.isl.DAP (partitioner/DAP.scala:53-64). The DAP schema clustering
// that produces <dataset>_local/class_index.txt_<N>_bal stays on the host (DAP.scala:116-354, 642-643);
// this tool runs what follows it: instance graph, instance partitioning, vertical partitions, indexes and
// statistics (DAP.scala:75, 85-94) on the GPU, and writes the reference's directory layout.
// Before that it writes the instance-scale inputs of the clustering (DAP.scala:82, Preprocessor.computeMeasures:
// schema_cc, schema_node_freq.txt, schema_node_count.txt) where they are missing; they need no class index, so a
// first call without one produces them and stops, the clustering runs on the host, and a second call partitions.
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "diaeresis_hip.h"

static bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

int main(int argc, char** argv) {
  if (argc - 1 != 5) {  // DAP.scala:53-57
    printf("-->Missing Arguments\n");
    printf("-->Arguments: dataset name, number of partitions, hdfs base path, insances path, schema path\n");
    return 255;  // System.exit(-1)
  }
  const std::string dataset = argv[1];
  const int n = atoi(argv[2]);
  std::string base = argv[3];
  const std::string instances = argv[4];
  if (!base.empty() && base.back() != '/') base += '/';  // DAP.scala:67-68
  const auto t0 = std::chrono::steady_clock::now();

  const std::string class_index = dataset + "_local/class_index.txt_" + std::to_string(n) + "_bal";
  const bool have_measures = exists(base + dataset + "_data/schema_cc") && exists(dataset + "_local/schema_node_freq.txt") &&
                             exists(dataset + "_local/schema_node_count.txt");
  if (!have_measures && !exists(class_index)) {
    dx_ctx* ctx = dx_init(0);
    if (!ctx) { fprintf(stderr, "dx-dap: no HIP device (this tool has no CPU path)\n"); return 1; }
    dx_part_stats st;
    dx_measure_sizes sz;
    int rc = dx_partition_file(ctx, instances.c_str(), nullptr, 1, &st);
    if (rc == DX_OK) rc = dx_measures_compute(ctx, &sz);
    if (rc == DX_OK) rc = dx_measures_write(ctx, base.c_str(), dataset.c_str(), "");
    if (rc != DX_OK) {
      fprintf(stderr, "dx-dap: error %d: %s\n", rc, dx_last_error(ctx));
      dx_free(ctx);
      return 1;
    }
    printf("dx-dap: measures over %llu triples: %llu classes, %llu (subType, pred, objType) groups; GPU time %.3f ms\n",
           (unsigned long long)st.n_triples, (unsigned long long)sz.n_classes, (unsigned long long)sz.n_cc, sz.device_ms);
    dx_free(ctx);
  }
  if (!exists(class_index)) {
    fprintf(stderr, "dx-dap: %s not found: the DAP schema clustering (DAP.clusterSchema) runs on the host and must "
                    "have written it before the instances can be partitioned\n", class_index.c_str());
    return 1;
  }
  const std::string clusters = base + dataset + "_data/cluster/clusters_" + std::to_string(n) + "_bal";
  if (exists(clusters)) {  // DAP.scala:86: the stage is skipped when its output exists
    printf("dx-dap: %s exists, nothing to do\n", clusters.c_str());
  } else {
    dx_ctx* ctx = dx_init(0);
    if (!ctx) { fprintf(stderr, "dx-dap: no HIP device (this tool has no CPU path)\n"); return 1; }
    dx_part_stats st;
    int rc = dx_partition_file(ctx, instances.c_str(), class_index.c_str(), n, &st);
    if (rc == DX_OK && !have_measures) {  // the same triples serve the measures
      rc = dx_measures_compute(ctx, nullptr);
      if (rc == DX_OK) rc = dx_measures_write(ctx, base.c_str(), dataset.c_str(), "");
    }
    if (rc == DX_OK) rc = dx_part_write_layout(ctx, base.c_str(), dataset.c_str(), "");
    if (rc != DX_OK) {
      fprintf(stderr, "dx-dap: error %d: %s\n", rc, dx_last_error(ctx));
      dx_free(ctx);
      return 1;
    }
    printf("dx-dap: %llu triples, %llu terms, %llu rows in %d partitions, %u tables; GPU time %.3f ms (%.1f M triples/s)\n",
           (unsigned long long)st.n_triples, (unsigned long long)st.n_terms, (unsigned long long)st.n_rows, n, st.n_tables,
           st.ms_total, st.n_triples / st.ms_total / 1e3);
    dx_free(ctx);
  }
  const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  printf("DAP - partitioning time: %lldms\n", (long long)ms);  // DAP.scala:96-97
  return 0;
}
