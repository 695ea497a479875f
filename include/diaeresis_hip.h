/* libdiaeresis_hip.so -- C ABI of the MI355X-native engine behind DIAERESIS' two hot paths.
 *
 * The reference (Scala on Spark) has no plugin or FFI seam: its partitioner and query processor
 * call Spark directly. Each entry point below replaces a group of those call sites; the comment
 * above it names them (paths relative to src/main/scala/org/ics/isl/ of the reference). A JVM host
 * binds these with JNI or Panama; the repository's own CLIs and Python wrapper use them as they are.
 *
 * Conventions: one dx_ctx per GPU and per host thread (the reference's driver is single-threaded);
 * calls block until the GPU work is done; 0 = success, negative = error (errno-style, below) with a
 * message in dx_last_error(); plain pointers and sizes only; the library owns every device buffer
 * except the ones explicitly passed in.
 */
#ifndef DIAERESIS_HIP_H
#define DIAERESIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DX_OK 0
#define DX_E_NOENT (-2)         /* table / file / term not found */
#define DX_E_IO (-5)            /* read or write failed */
#define DX_E_NOMEM (-12)        /* device or host allocation failed */
#define DX_E_INVAL (-22)        /* bad argument */
#define DX_E_RANGE (-34)        /* input exceeds a documented limit */
#define DX_E_MALFORMED (-74)    /* input line or SQL text the reference would throw on */
#define DX_E_UNSUPPORTED (-95)  /* valid for the reference but outside this engine's subset */
#define DX_E_HIP (-1000)        /* HIP runtime error; text in dx_last_error */

#define DX_MAX_PARTITIONS 64

typedef struct dx_ctx dx_ctx;
typedef struct dx_result dx_result;

/* ---- context -------------------------------------------------------------------------------
 * Replaces SparkSession.builder...getOrCreate() (partitioner/DAP.scala:41-48,
 * query_processor/QueryProcessor.scala:116-154) and spark.stop() (DAP.scala:98, QueryProcessor.scala:109). */
dx_ctx* dx_init(int device);
void dx_free(dx_ctx* ctx);
const char* dx_last_error(dx_ctx* ctx);

/* ---- partitioner: the post-DAP stage ----------------------------------------------------------- */
typedef struct dx_part_stats {
  uint64_t n_bytes;         /* input text bytes */
  uint64_t n_triples;       /* input records = lines not starting with '#' (DAP.scala:83 counts all lines) */
  uint64_t n_unique;        /* distinct (s, p, o) */
  uint64_t n_terms;         /* dictionary size */
  uint64_t n_vertices;      /* Preprocessor.scala:131-143 */
  uint64_t n_edges;         /* Preprocessor.scala:153-167 */
  uint64_t n_rows;          /* rows over all first-level partitions (replicas counted) */
  uint64_t n_node_index;    /* rows of node_index_N */
  uint64_t n_vp_index;      /* (uri, role, predicate) entries over all partitions */
  uint32_t n_predicates;
  uint32_t n_tables;        /* non-empty (partition, predicate) vertical-partition tables */
  uint32_t hash_retries;    /* dictionary rebuilds caused by a true 64-bit hash collision */
  uint32_t n_dict_entries;  /* occurrences that went to the dictionary table: the rest equalled an earlier one of their 32 KiB tile */
  double ms_total;          /* device time of the whole call, inputs already in HBM */
  double ms_stage[12];      /* see dx_part_stage_name() */
} dx_part_stats;

/* Replaces, in one call, Preprocessor.createInstanceGraph (Preprocessor.scala:116-192, minus the
 * parquet writes), Loader.loadInstanceGraph (Loader.scala:19-38), Loader.loadClassIndexBal
 * (Loader.scala:114-123), DAP.partitionInstances (partitioner/DAP.scala:567-629),
 * DAP.partitionClusters (:360-382), DAP.subPartitionClusters (:385-509) and
 * DAP.computePartitionStatistics (:101-114), for N-Triples text that is already resident in HBM.
 * d_text: device pointer to n_bytes of UTF-8 text. class_index_text: contents of
 * <dataset>_local/class_index.txt_<N>_bal (host string, NUL-terminated). Results stay in HBM inside
 * ctx (replacing any earlier run) until dx_free; read them with the dx_part_* accessors, register
 * them for queries with dx_load_table, or persist them with dx_part_write_layout.
 * Limits per GPU (DX_E_RANGE beyond them): fewer than 2^32/3 input triples, fewer than 2^32 dictionary terms, and
 * fewer than 2^32 rows summed over the first-level partitions (a triple placed in k partitions is k rows). */
int dx_partition_device(dx_ctx* ctx, const void* d_text, uint64_t n_bytes, const char* class_index_text,
                        int n_partitions, dx_part_stats* stats);
/* Same, reading the instances file and the class index file from the local file system. */
int dx_partition_file(dx_ctx* ctx, const char* instances_path, const char* class_index_path, int n_partitions,
                      dx_part_stats* stats);
const char* dx_part_stage_name(int i);

/* Dictionary (replaces the strings Spark carries through every stage; ids are int64, dense, assigned
 * in order of first occurrence in the input). */
int64_t dx_term_count(dx_ctx* ctx);
/* Sum of the byte lengths of the given terms (to size the buffer of dx_terms_decode). id < 0 or >= count -> DX_E_RANGE. */
int dx_terms_bytes(dx_ctx* ctx, const int64_t* ids, uint64_t n, uint64_t* total_bytes);
/* Concatenated UTF-8 bytes of the terms; offsets[n+1] delimit them. */
int dx_terms_decode(dx_ctx* ctx, const int64_t* ids, uint64_t n, char* bytes, uint64_t cap, uint64_t* offsets);
/* ids[i] = id of the i-th string, or -1 when it is not in the dictionary. */
int dx_terms_lookup(dx_ctx* ctx, const char* bytes, const uint64_t* offsets, uint64_t n, int64_t* ids);

/* First-level partitions (DAP.scala:364-367: one part file per partition, columns _1=s,_2=o,_3=p). */
int dx_part_count(dx_ctx* ctx);
int64_t dx_part_rows(dx_ctx* ctx, int partition);
/* Copies the partition's rows to host arrays (each of dx_part_rows entries); p holds predicate term ids. */
int dx_part_fetch(dx_ctx* ctx, int partition, int64_t* s, int64_t* o, int64_t* p);

/* Vertical-partition tables (DAP.scala:419-437). Table i is (partition, predicate term). */
typedef struct dx_table_info {
  int32_t partition;
  int32_t in_layout;        /* 0 when the predicate has >= 150 UTF-16 units and is left out of the VP dirs (DAP.scala:434) */
  int64_t predicate;        /* term id */
  int64_t rows;
} dx_table_info;
int dx_table_count(dx_ctx* ctx);
int dx_table_info_at(dx_ctx* ctx, int i, dx_table_info* out);
int dx_table_fetch(dx_ctx* ctx, int i, int64_t* s, int64_t* o);

/* node_index_N (DAP.scala:402-412): distinct (uri, type) pairs. */
int64_t dx_node_index_rows(dx_ctx* ctx);
int dx_node_index_fetch(dx_ctx* ctx, int64_t* uri, int64_t* type);

/* classSubPart_index_N__0000i (DAP.scala:450-501), as (uri, role, predicate) with role 0 = 's', 1 = 'o'. */
int64_t dx_vp_index_rows(dx_ctx* ctx, int partition);
int dx_vp_index_fetch(dx_ctx* ctx, int partition, int64_t* uri, int32_t* role, int64_t* predicate);

/* Instance graph of Preprocessor.scala:142-167: vertex term ids in String.compareTo order (position =
 * the reference's vertex id) and the distinct edges as (src position, dst position, predicate term id). */
int dx_instance_graph_sizes(dx_ctx* ctx, int64_t* n_vertices, int64_t* n_edges);
int dx_instance_graph_fetch(dx_ctx* ctx, int64_t* vertex_terms, int64_t* edge_src, int64_t* edge_dst, int64_t* edge_pred);

/* Instance-scale inputs of the schema clustering, computed from the state dx_partition* left on the GPU
 * (replaces Preprocessor.computeCC :230-264, computeSchemaNodeFreq :312-326 and computeSchemaNodeCount
 * :328-339, called from DAP.main through Preprocessor.computeMeasures, DAP.scala:82). Type lists keep
 * their multiplicity, as the reference's Seq[String] does: a type triple stated twice counts twice.
 *   dx_measures_compute   runs the kernels; n_classes = classes that type a subject, n_cc = (subType, pred,
 *                         objType) groups; device_ms = GPU time of the call.
 *   dx_measures_classes   per class (ascending term id): the class, its schema_node_freq.txt value (occurrences
 *                         in the type lists of distinct typed endpoints of edges; 0 = no line in the file) and its
 *                         schema_node_count.txt value (occurrences over all edges with a typed endpoint).
 *   dx_measures_cc        per group, ordered by (subType, pred, objType) in class / predicate order: rows and
 *                         distinct object instances (the two numbers of a schema_cc line).
 *   dx_measures_write     <base><dataset>_data/schema_cc/part-00000 (+ _SUCCESS) with the text lines
 *                         "((subType,pred,objType),rows,dist)" that Loader.loadCC parses (Loader.scala:318-322),
 *                         <local_dir><dataset>_local/schema_node_freq.txt and schema_node_count.txt
 *                         ("class\tvalue", descending value; Constatnts.scala:13,34,40). Each of the three is
 *                         written only if it does not exist yet, as Preprocessor.computeMeasures does (:37-46).
 * With a communicator set (several GPUs) dx_measures_compute is a collective call: every rank gathers all ranks' own
 * triples and computes the same result over the whole input.
 * They need no class index: dx_partition_file takes a null class_index_path (no class is placed anywhere), which is
 * how dx-dap produces them before the schema clustering has run. */
typedef struct dx_measure_sizes {
  uint64_t n_classes, n_cc, n_type_entries, n_cc_rows;
  double device_ms;
} dx_measure_sizes;
int dx_measures_compute(dx_ctx* ctx, dx_measure_sizes* out);
int dx_measures_classes(dx_ctx* ctx, int64_t* class_term, int64_t* node_freq, int64_t* node_count);
int dx_measures_cc(dx_ctx* ctx, int64_t* sub_type, int64_t* pred, int64_t* obj_type, int64_t* rows, int64_t* dist);
int dx_measures_write(dx_ctx* ctx, const char* base, const char* dataset, const char* local_dir);

/* Writes what DAP.main leaves behind (DAP.scala:85-94; paths of Constatnts.scala:9-40), as Parquet files with the
 * Scala tools' schemas (utils/Types.scala:7-16): the first-level partitions (_1,_2,_3; DAP.scala:364-367), the vertical
 * partition directories 0000i/_3=<predicate> (_1,_2; :433-437), node_index_<N> (uri, uriType; :405-412),
 * classSubPart_index_<N>__0000i (uri, parts; :496-499), _SUCCESS files, and <dataset>_local/statistics.txt_<N>,
 * partition_stat.txt_<N>_bal, statTables.txt. Beside them, under <base><dataset>_data/dx_<N>/ (a directory the Scala
 * tools never list), the dictionary and the tables' id columns for dx_open_dataset / dx_load_table.
 * DX_E_INVAL if the clusters directory exists (the reference skips the stage then, DAP.scala:86). */
int dx_part_write_layout(dx_ctx* ctx, const char* base, const char* dataset, const char* local_dir);
/* Where the time of the last dx_part_write_layout went, in the order: whole call, directories and text files, page
 * encoding on the GPU, the files' bytes coming to pinned host memory, first write started .. last write done, time summed
 * over the writing threads (all in ms), bytes written, writing threads. Fills min(n, 8) values. The reference only has
 * its one stage timer for this (DAP.scala:73, 96-97). */
int dx_layout_times(dx_ctx* ctx, double* out, int n);

/* ---- query processor ------------------------------------------------------------------------------ */
/* Points the context at a layout (replaces the path arithmetic of QueryProcessor.loadDataset,
 * QueryProcessor.scala:308-325). A layout written by dx_part_write_layout carries its dictionary and id columns under
 * <dataset>_data/dx_<N>/: the dictionary is loaded from there and tables come in one by one with dx_load_table. A layout
 * without that directory (written by the Scala tools) is brought in through dx_import_layout. */
int dx_open_dataset(dx_ctx* ctx, const char* base, const char* dataset, int n_partitions);
/* Reads every vertical-partition table of a layout from its Parquet files,
 * <base><dataset>_data/cluster/clusters_<N>_bal/<PPPPP>/_3=<predicate>/part-*.parquet with the string columns _1, _2
 * (DAP.scala:433-437): what spark.read.load(...).as[(String, String)] does table by table in
 * QueryProcessor.loadDataset (QueryProcessor.scala:322-333), here for all tables at once, because one dictionary is
 * built over their strings. Reads what parquet-mr (Spark) and dx_part_write_layout write: REQUIRED or OPTIONAL
 * BYTE_ARRAY columns without nulls, PLAIN and dictionary encodings, data pages of both versions, uncompressed or
 * snappy. Afterwards the tables are resident as id columns and dx_load_table / dx_unload_all hand them out and take
 * them back by key without reading anything again. n_tables / n_rows (may be null): what was read. */
int dx_import_layout(dx_ctx* ctx, const char* base, const char* dataset, int n_partitions, uint64_t* n_tables, uint64_t* n_rows);
/* Replaces QueryProcessor.loadDataset + preloadTables (QueryProcessor.scala:210-236, 308-339) for one
 * entry of the `partitions` line ("00001-_3=<sanitised predicate>"): makes the table resident in HBM and
 * visible to SQL under "table" + key with '-' -> '_' and '=' -> '_E_' (:216). Memoised like dataMap (:327-337). */
int dx_load_table(dx_ctx* ctx, const char* partition_key, uint64_t* rows);
/* The whole `partitions` line of a translated query at once, as preloadTables takes it (QueryProcessor.scala:210-236):
 * the tables' statistics for the planner are then gathered behind one wait instead of one per table.
 * rows_total = sum of the rows (the reference's Load_data, duplicates in the list counted each time). */
int dx_load_tables(dx_ctx* ctx, const char* const* partition_keys, int n_keys, uint64_t* rows_total);
/* For tables that reach the host as strings (Parquet written by the Scala tools, decoded by the host's
 * Parquet library): installs a dictionary of n_terms distinct terms (term i = bytes[offsets[i], offsets[i+1]),
 * its id is i), replacing any dictionary and every loaded table, then tables are registered by id.
 * Together they replace spark.read.load(...).as[(String, String)] in QueryProcessor.loadDataset
 * (QueryProcessor.scala:322-333). */
int dx_dictionary_load(dx_ctx* ctx, const char* bytes, const uint64_t* offsets, uint64_t n_terms);
int dx_load_table_ids(dx_ctx* ctx, const char* partition_key, const int64_t* s, const int64_t* o, uint64_t rows);
/* Replaces unloadTables (QueryProcessor.scala:239-251). */
int dx_unload_all(dx_ctx* ctx);

#define DX_QUERY_COUNT 0        /* what the warm path times: result.count (QueryProcessor.scala:173-176) */
#define DX_QUERY_BINDINGS 1     /* materialise bindings (QueryProcessorCold.scala:119-127) */
/* Replaces executeFinalQuery -> spark.sqlContext.sql(query) (QueryProcessor.scala:184-195) followed by
 * result.count. sql is one statement of the translated-query grammar (QueryTranslator.scala:363-392, 417-543). */
int dx_query(dx_ctx* ctx, const char* sql, int mode, dx_result** out);
/* The query map of a translated file that holds several statements (QueryProcessor.scala:73-88 reads them, :184-195
 * executes each and folds the results with customJoin, :197-208: an inner join on the output columns two results share,
 * commonColumns :276-278, or a cross join when they share none). Same result handle as dx_query: the count of the joined
 * result (what the reference's timed result.count returns), or its bindings; columns in Spark's order for such a join
 * (shared, left, right). The statements are joined in the order given. With n == 1 this is dx_query. */
int dx_query_joined(dx_ctx* ctx, const char* const* statements, int n, int mode, dx_result** out);
/* No counterpart in the reference. The count of dx_query(sql, DX_QUERY_COUNT) for callers that cannot allocate
 * device memory while they run: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc. The statement's device scratch (resolved constants, row counters, intermediate relations, hash-join
 * heads and chains) comes out of one fixed 32 MiB array that every GPU the library runs on holds from load time
 * until the process ends; the call makes no hipMalloc / hipFree. A statement that needs more returns DX_E_RANGE
 * (use dx_query). The tables must have been loaded with dx_load_table / dx_load_tables (their statistics are
 * gathered there; DX_E_INVAL otherwise); a UNION over a directory shared by several predicates must have gone
 * through dx_query once. One GPU only (DX_E_UNSUPPORTED with a communicator set). All contexts of a GPU share
 * the array: calls that overlap in time on one GPU race, so serialise them. */
int dx_query_count_noalloc(dx_ctx* ctx, const char* sql, uint64_t* count);
/* The context's device allocator: buffers it has handed out since dx_init (from its cache or from hipMalloc) and the bytes
 * of those still in use. No counterpart in the reference; for callers of dx_query_count_noalloc who want to check, as this
 * repository's tests do, that a call left both numbers as they were. Either pointer may be NULL. */
int dx_allocator_stats(dx_ctx* ctx, uint64_t* requests, uint64_t* live_bytes);
uint64_t dx_result_count(dx_result* r);
int dx_result_columns(dx_result* r);
const char* dx_result_column_name(dx_result* r, int i);
/* Rows as term ids, column-major (column i at ids + i * count), sorted by the decoded strings of the
 * columns in name order when sorted != 0. Needs DX_QUERY_BINDINGS. */
int dx_result_fetch_ids(dx_result* r, int64_t* ids, int sorted);
/* Ids <= -2 in a result are constants projected by the statement itself (`'<...>' AS P`,
 * QueryTranslator.scala:485-526), which are not dictionary terms; this returns their text, or NULL. */
const char* dx_result_virtual_term(dx_result* r, int64_t id);
/* The executed plan as text, one line per step: tables, filters, rows after the step (what the reference
 * gets from result.explain, QueryProcessor.scala:178). */
const char* dx_result_plan(dx_result* r);
/* Device time of the statement in milliseconds (hipEvents around the launches). */
double dx_result_device_ms(dx_result* r);
/* A result holds device memory of its context: free it before dx_free of that context. */
void dx_result_free(dx_result* r);

/* ---- query processor on several GPUs ------------------------------------------------------------------
 * One process and one dx_ctx per GPU. Every rank opens the same layout (one dictionary, so term ids agree),
 * loads the tables of the first-level partitions it owns with dx_load_table and names the others with
 * dx_declare_remote_table (dx_load_tables_sharded does both). dx_query is then collective: every rank passes the
 * same statement. A statement whose tables all lie in first-level partitions of one rank (partition q belongs to
 * rank q % world; the partition is read off the table names) is answered by that rank alone, and one sum brings the
 * count to the others. Otherwise scans and filters run where the table lives, and before each join the small side is
 * sent to all ranks or both inputs are redistributed by a hash of the join key. This is the exchange Spark plans
 * inside spark.sqlContext.sql (QueryProcessor.scala:187,192, spark.sql.shuffle.partitions :124,139) and its broadcast
 * of a small side, with the GPUs of the node in the place of its shuffle partitions.
 * The library asks five collectives of a communicator (struct dx_comm). dx_comm_rccl_init makes one over RCCL
 * inside the library (production: xGMI between the GPUs of a node); a host may hand in its own instead with
 * dx_set_comm (the tests run two ranks over gloo that way). Buffers named *_dev are device memory of this rank's
 * GPU. Counts are in elements, grouped by peer in rank order. */
typedef struct dx_comm {
  int rank, world;
  void* user;
  /* all-to-all of small host vectors: send and recv hold `world` groups of n words; recv group p = the group
   * peer p passed at position `rank`. n <= 16. Returns when recv is complete. */
  int (*exchange_host)(void* user, const uint64_t* send, uint64_t* recv, int n);
  /* all-to-all with per-peer counts of elements of elem_bytes bytes */
  int (*exchange)(void* user, const void* send_dev, const uint64_t* send_counts, void* recv_dev, const uint64_t* recv_counts,
                  int elem_bytes);
  /* element-wise sum of n <= 1024 host values over all ranks, result on every rank */
  int (*allreduce_sum)(void* user, uint64_t* values, int n);
  /* recv_dev = the ranks' send buffers one after another in rank order (recv_counts[p] elements from p). Needed by the
   * partitioner and by the queries (a small join side and a cross product go to every rank through it): dx_set_comm
   * refuses a group of several ranks without it. */
  int (*allgather)(void* user, const void* send_dev, uint64_t send_count, void* recv_dev, const uint64_t* recv_counts, int elem_bytes);
  /* element-wise sum of a device array of n elements of 1 or 8 bytes, in place, result on every rank; the partitioner's
   * and the measures' (may be NULL when only queries run on several GPUs) */
  int (*allreduce_sum_dev)(void* user, void* dev, uint64_t n, int elem_bytes);
  /* 0: the three device collectives return when their result is complete, and the library waits for its own
   *    stream before it calls them.
   * 1: they are queued on the context's stream like kernels (the library neither waits before nor assumes
   *    completion after; it synchronises the stream where it needs a result on the host). */
  int stream_ordered;
} dx_comm;
/* comm == NULL or world == 1 returns the context to single-GPU execution.
 * With a communicator that has all five collectives, dx_partition_device / dx_partition_file are collective
 * too: every rank passes its own part of the input text (cut at line ends) and the same class index; the
 * ranks agree on one dictionary (same ids everywhere) and rank r ends up holding the first-level
 * partitions q with q % world == r of the whole input, as partitionClusters' CustomPartitioner routes
 * them (DAP.scala:541-551), with their VP tables and VP index, and its share of the node index. */
int dx_set_comm(dx_ctx* ctx, const dx_comm* comm);
/* The communicator over RCCL, made inside the library (replaces the executors' shuffle service Spark brings up
 * with the session, QueryProcessor.scala:116-154, DAP.scala:41-48). Rank 0 makes a unique id and hands its
 * DX_RCCL_ID_BYTES bytes to the other ranks by any means; then every rank calls dx_comm_rccl_init with its own
 * context (one GPU per rank; the call returns when all `world` ranks have made it, or with DX_E_IO when they have not
 * all arrived within DX_RCCL_INIT_TIMEOUT_MS, 10 minutes by default: the process should then end). dx_comm_rccl_init_file
 * does both over a path all ranks can reach: rank 0 removes what lies there, writes the id, and removes the file again
 * once the group has formed; the others wait up to timeout_ms for the file, and every rank up to timeout_ms for the
 * group. dx_comm_rccl_init_job is the same with the number of the job in the file's name (`path.<job in hex>`; any
 * number that the ranks of one launch share and other launches do not), so that a file which a launch that died has left
 * behind is never read; without it such a file can cost the rank that read it its timeout_ms.
 * The collectives are queued on the context's stream. dx_set_comm(ctx, NULL), another dx_comm_rccl_init and
 * dx_free leave the group. world == 1 is single-GPU execution. */
#define DX_RCCL_ID_BYTES 128
int dx_rccl_unique_id(unsigned char id[DX_RCCL_ID_BYTES]);
int dx_comm_rccl_init(dx_ctx* ctx, const unsigned char id[DX_RCCL_ID_BYTES], int rank, int world);
int dx_comm_rccl_init_file(dx_ctx* ctx, const char* path, int rank, int world, int timeout_ms);
int dx_comm_rccl_init_job(dx_ctx* ctx, const char* path, uint64_t job, int rank, int world, int timeout_ms);
/* Drives each of the communicator's collectives once with known patterns on small device buffers and checks
 * what arrives (a collective call; every rank checks its own share). Meant to be called once after the group has
 * been formed, before a long job relies on it. DX_E_IO with the first mismatch in dx_last_error. */
int dx_comm_selftest(dx_ctx* ctx);
/* preloadTables (QueryProcessor.scala:210-236) on several GPUs, a collective call: every rank passes the same
 * `partitions` line; the tables of first-level partition q are loaded on rank q % world (CustomPartitioner's
 * placement, DAP.scala:541-551) and declared remote elsewhere, and the owners' row and distinct counts are
 * shared, so that every rank plans alike. rows_total = rows loaded over all ranks (the reference's Load_data). */
int dx_load_tables_sharded(dx_ctx* ctx, const char* const* partition_keys, int n_keys, uint64_t* rows_total);
/* A table of the `partitions` line that another rank holds (same key syntax as dx_load_table). */
int dx_declare_remote_table(dx_ctx* ctx, const char* partition_key);
/* Rows, distinct subjects and distinct objects of a table: read them where it is loaded, sum or share them
 * between the ranks, and set them everywhere, so that every rank derives the same join order. */
int dx_table_stats(dx_ctx* ctx, const char* partition_key, uint64_t out[3]);
int dx_set_table_stats(dx_ctx* ctx, const char* partition_key, const uint64_t in[3]);
/* With a communicator set, dx_result_count is the count over all ranks and dx_result_fetch_ids returns this
 * rank's share of the bindings: dx_result_local_rows of them (unsorted: the caller merges the shares). */
uint64_t dx_result_local_rows(dx_result* r);

#ifdef __cplusplus
}
#endif
#endif
