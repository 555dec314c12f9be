// read_pc_mat_module -- pybind11 surface of the query library, same module name, functions, argument names
// and return shapes as the reference's src/bindings.cpp:110-126 (used by src/read_pc_mat.py:7-44):
//   query(matrix_folder, db_folder, query_file) -> list[dict{id, neighbor_ids: list[str],
//                                                            jaccard_similarities: np.float32[]}]
//   query_sliced(matrix_folder, db_folder, row_file, col_file)
//                                             -> dict{'row-list', 'col-list', 'jac-dict': {row_id: list[float]}}
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "read_pc_mat.hpp"

namespace py = pybind11;

// the array owns its buffer through a capsule (src/bindings.cpp:11-34)
template <typename T>
static py::array_t<T> vector_to_numpy(std::vector<T>&& vec) {
    auto* heap_vec = new std::vector<T>(std::move(vec));
    py::capsule free_when_done(heap_vec, [](void* p) { delete static_cast<std::vector<T>*>(p); });
    return py::array_t<T>({(py::ssize_t)heap_vec->size()}, {(py::ssize_t)sizeof(T)}, heap_vec->data(), free_when_done);
}

static py::list query_py(std::string matrix_folder, std::string db_folder, std::string query_file) {
    std::vector<std::string> identifiers;
    auto id_to_index = pc_mat::load_vector_identifiers(db_folder, identifiers);
    std::vector<std::string> query_ids_str;
    std::vector<int32_t> queries = pc_mat::read_queries_from_file(query_file, id_to_index, query_ids_str);
    std::vector<float> vector_norms;
    pc_mat::load_vector_norms(db_folder, vector_norms);
    std::vector<pc_mat::Result> results = pc_mat::query(matrix_folder, queries, vector_norms, identifiers);
    py::list all_results;
    for (auto& res : results) {
        py::dict res_dict;
        res_dict["id"] = res.self_id;
        py::list ids;
        for (const auto& s : res.neighbor_ids) ids.append(s);
        res_dict["neighbor_ids"] = ids;
        res_dict["jaccard_similarities"] = vector_to_numpy(std::move(res.jaccard_similarities));
        all_results.append(res_dict);
    }
    return all_results;
}

static py::dict query_sliced_py(std::string matrix_folder, std::string db_folder, std::string row_file,
                                std::string col_file) {
    std::vector<std::string> identifiers;
    auto id_to_index = pc_mat::load_vector_identifiers(db_folder, identifiers);
    std::vector<std::string> row_vec, col_vec;
    std::vector<int32_t> row_query_vec = pc_mat::read_queries_from_file(row_file, id_to_index, row_vec);
    std::vector<int32_t> col_query_vec = pc_mat::read_queries_from_file(col_file, id_to_index, col_vec);
    const int total_vectors = (int)identifiers.size();
    std::vector<float> vector_norms;
    pc_mat::load_vector_norms(db_folder, vector_norms);
    std::vector<std::vector<float>> results =
        pc_mat::query_sliced(matrix_folder, row_query_vec, col_query_vec, total_vectors, vector_norms);
    py::list row_list, col_list;
    for (const auto& row : row_vec) row_list.append(row);
    for (const auto& col : col_vec) col_list.append(col);
    py::dict jaccard_dict;
    for (size_t i = 0; i < results.size(); i++) {
        py::list jaccard_list;
        for (float v : results[i]) jaccard_list.append(v);
        jaccard_dict[row_vec[i].c_str()] = jaccard_list;
    }
    py::dict final_result;
    final_result["row-list"] = row_list;
    final_result["col-list"] = col_list;
    final_result["jac-dict"] = jaccard_dict;
    return final_result;
}

PYBIND11_MODULE(read_pc_mat_module, m) {
    m.doc() = "Module for querying pairwise comparison matrix";
    m.def("query", &query_py, py::arg("matrix_folder"), py::arg("db_folder"), py::arg("query_file"),
          "Compute neighbors for queries in the given matrix folder, database folder and query file / ids;"
          " returns a list of dictionaries with neighbor IDs and jaccard similarities.");
    m.def("query_sliced", &query_sliced_py, py::arg("matrix_folder"), py::arg("db_folder"), py::arg("row_file"),
          py::arg("col_file"),
          "Compute neighbors for queries in the given matrix folder, database folder and from the corresponding row-col "
          "files; returns a dictionary containing row, col IDS and their corresponding jaccard similarities.");
}
