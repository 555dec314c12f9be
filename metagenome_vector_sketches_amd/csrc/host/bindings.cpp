// read_pc_mat_module -- Python surface of the shard reader.  The module name, the two function names, their
// keyword arguments and the shapes they return are the reference's (src/bindings.cpp:110-126; consumer
// src/read_pc_mat.py:7-44):
//   query(matrix_folder, db_folder, query_file)
//       -> [ {'id': str, 'neighbor_ids': [str, ...], 'jaccard_similarities': float32 ndarray}, ... ]
//   query_sliced(matrix_folder, db_folder, row_file, col_file)
//       -> {'row-list': [str], 'col-list': [str], 'jac-dict': {row id: [float, ...]}}
// Everything behind that is this build's: one SampleTable scan of vector_norms.txt per call, the MatrixView row
// iterator, results written straight into Python objects (numpy allocates the arrays; nothing is handed over
// through capsules).
#include <cstring>

#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include "read_pc_mat.hpp"

namespace py = pybind11;

namespace {

// the rows named in `file`, with the text they were named by
struct Request {
    std::vector<int> rows;
    std::vector<std::string> labels;
    Request(const std::string& file, const pc_mat::SampleTable& db) {
        rows = pc_mat::read_queries_from_file(file, db.index_of, labels);
    }
};

pc_mat::SampleTable open_db(const std::string& db_folder) {
    pc_mat::SampleTable db(db_folder);
    if (!db.opened) throw std::runtime_error("cannot read " + pc_mat::SampleTable::path_in(db_folder));
    return db;
}

py::list strings(const std::vector<std::string>& v) {
    py::list out(v.size());
    for (size_t i = 0; i < v.size(); ++i) out[i] = py::str(v[i]);
    return out;
}

py::list neighbours(const std::string& matrix_folder, const std::string& db_folder, const std::string& query_file) {
    pc_mat::SampleTable db = open_db(db_folder);
    Request req(query_file, db);
    std::vector<pc_mat::Result> found = pc_mat::query(matrix_folder, req.rows, db.norms, db.names);
    py::list out(found.size());
    for (size_t i = 0; i < found.size(); ++i) {
        const pc_mat::Result& r = found[i];
        py::array_t<float> jac((py::ssize_t)r.jaccard_similarities.size());
        if (!r.jaccard_similarities.empty())
            std::memcpy(jac.mutable_data(), r.jaccard_similarities.data(), r.jaccard_similarities.size() * sizeof(float));
        py::dict d;
        d["id"] = py::str(r.self_id);
        d["neighbor_ids"] = strings(r.neighbor_ids);
        d["jaccard_similarities"] = std::move(jac);
        out[i] = std::move(d);
    }
    return out;
}

py::dict slice(const std::string& matrix_folder, const std::string& db_folder, const std::string& row_file,
               const std::string& col_file) {
    pc_mat::SampleTable db = open_db(db_folder);
    Request rows(row_file, db), cols(col_file, db);
    const std::vector<std::vector<float>> values =
        pc_mat::query_sliced(matrix_folder, rows.rows, cols.rows, (int32_t)db.names.size(), db.norms);
    py::dict per_row;
    for (size_t i = 0; i < values.size(); ++i) {
        py::list line(values[i].size());
        for (size_t j = 0; j < values[i].size(); ++j) line[j] = py::float_(values[i][j]);
        per_row[py::str(rows.labels[i])] = std::move(line);
    }
    py::dict out;
    out["row-list"] = strings(rows.labels);
    out["col-list"] = strings(cols.labels);
    out["jac-dict"] = std::move(per_row);
    return out;
}

}  // namespace

PYBIND11_MODULE(read_pc_mat_module, m) {
    m.doc() = "Reader for the pairwise-comparison matrix shards written by pairwise_comp_optimized";
    m.def("query", &neighbours, py::arg("matrix_folder"), py::arg("db_folder"), py::arg("query_file"),
          "Neighbours of the samples listed in query_file (names or row indices, one per line): a list with one dict "
          "per query -- 'id', 'neighbor_ids' (strongest first) and 'jaccard_similarities' (float32 array).");
    m.def("query_sliced", &slice, py::arg("matrix_folder"), py::arg("db_folder"), py::arg("row_file"),
          py::arg("col_file"),
          "Rows x columns slice of the matrix for the samples listed in row_file / col_file: a dict with 'row-list', "
          "'col-list' and 'jac-dict' (row id -> list of Jaccard estimates in column order; 0 where nothing is stored).");
}
